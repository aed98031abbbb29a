#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04h_gputests.txt 2>&1; echo "gpu tests rc $?"
tail -8 gpurun_out/r04h_gputests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
