#!/bin/bash
# usage: tools/ab_build.sh  -- variants/b_work.so = the working tree's library, variants/a_head.so = HEAD's (for tools/ab.sh)
set -e
mkdir -p variants
python -m boardlaw_amd.build > /dev/null 2>&1; cp boardlaw_amd/libboardlaw_amd.so variants/b_work.so
git stash -q; python -m boardlaw_amd.build > /dev/null 2>&1; cp boardlaw_amd/libboardlaw_amd.so variants/a_head.so; git stash pop -q
python -m boardlaw_amd.build > /dev/null 2>&1
ls -la variants
