#!/bin/bash
# usage (GPU box, repo root): tools/hex_tile_pmc.sh <tag>  -> gpurun_out/<tag>/hex_pmc_*.csv: SQ / traffic counters of the board kernels at 2^20 envs
tag=${1:-hexpmc}; out=$PWD/gpurun_out/$tag; mkdir -p $out
repo=$PWD; cd /tmp; export TMPDIR=/tmp
cat > /tmp/hex_once.py <<PY
import sys; sys.path.insert(0, '$repo')
import torch
from boardlaw_amd import _native
from boardlaw_amd.hex import Hex
S, B = 11, 1 << 20
gen = torch.Generator(device='cuda'); gen.manual_seed(7)
w = Hex.initial(B, S)
for _ in range(S * S // 3):
    v = w.valid
    w, _ = w.step((torch.rand(v.shape, device='cuda', generator=gen) * v).argmax(-1), check=False)
v = w.valid
a = (torch.rand(v.shape, device='cuda', generator=gen) * v).argmax(-1)
for _ in range(5):
    w2, _ = w.step(a, check=False); w2.valid
    sc = w.board.clone(); from boardlaw_amd.hex import cuda as hc; hc.step(sc, w.seats, a.int())
torch.cuda.synchronize()
PY
run() { name=$1; shift; rm -rf /tmp/hp_$name
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/hp_$name --output-format csv -- python /tmp/hex_once.py > /dev/null 2> $out/hex_pmc_$name.err
  python $repo/tools/pmc_summary.py /tmp/hp_$name hex_ > $out/hex_pmc_$name.csv; cat $out/hex_pmc_$name.csv | cut -c1-260; }
run SQ1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run SQ2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run FETCH FETCH_SIZE
run WRITE WRITE_SIZE
