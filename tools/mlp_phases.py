"""Shader-clock phases of bl_sim_infer_finish (workgroup 0, thread 0) inside a real search: build the library with -DBL_MLP_CLK
into tools/micro/libboardlaw_clk.so (tools/mlp_phases.py --build, where hipcc is), then run this on the GPU box.
MLP_ENVS=N picks the batch (default 4096); BL_MLP_ROWS=32|64 the rows per workgroup (default: the library's choice by the batch;
with 64 the finish epilogue runs two passes of four envs per wave, the second pass's phases are listed separately)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'tools', 'micro', 'libboardlaw_clk.so')
if '--build' in sys.argv:
    sys.path.insert(0, ROOT)
    from boardlaw_amd import build as b
    subprocess.check_call([b.shutil.which('hipcc') or '/opt/rocm/bin/hipcc'] + b.FLAGS + ['-shared', '-DBL_MLP_CLK'] + b.SOURCES + ['-o', LIB])
    sys.exit(0)
import numpy as np, torch
sys.path.insert(0, ROOT)
from boardlaw_amd import _native
_native.LIBPATH = LIB
from boardlaw_amd import networks
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTS, MoveRng
from bench import premix
gen = torch.Generator(device='cuda'); gen.manual_seed(0); torch.manual_seed(0)
ENVS = int(os.environ.get('MLP_ENVS', 4096))
worlds = premix(Hex.initial(ENVS, 9), 27, gen)
net = networks.Inference(networks.FCModel(worlds.obs_space, worlds.action_space, 512, 4).cuda(), fused=True)
net.refresh()
m = MCTS(worlds, n_nodes=64, rng=MoveRng(), obs_half=True)
m.rng.start(63, m.decisions.logits[:, :, 0])
m.initialize(net)
names = {50: 'obs loads issued', 51: 'weight prefetch issued', 52: 'obs in LDS', 53: 'staging barrier', 1: 'finish prefetch issued'}
for l in range(5): names.update({2 + 3 * l: f'layer{l} gemm', 3 + 3 * l: f'layer{l} epilogue', 4 + 3 * l: f'layer{l} barrier'})
names.update({34: 'heads gemm + staging', 35: 'logits/v stores, tanh', 36: 'compacted row', 37: 'backup scan', 38: 'w/n stores + q range', 40: 'atomics, end'})
second = {41: 'pass 2: Out -> regs, masks', 42: 'pass 2: max butterfly', 43: 'pass 2: exp', 44: 'pass 2: sum butterfly', 45: 'pass 2: log, logit bits', 46: 'pass 2: exp-table gathers', 20: 'pass 2: logits/v stores, tanh', 21: 'pass 2: compacted row', 22: 'pass 2: backup scan', 23: 'pass 2: w/n stores + q range'}
tot = {}
for sim in range(1, 64):
    m.simulate(net)
    if sim in (5, 30, 60):
        torch.cuda.synchronize()
        clk = np.zeros(64, np.int64); _native.lib().bl_mlp_debug_clk(ctypes.c_void_p(clk.ctypes.data))
        print(f'--- sim {sim}: workgroup 0 total {clk[40] - clk[0]} cycles')
        prev = clk[0]
        names.update({54: 'Out -> regs, masks', 55: 'max butterfly', 56: 'exp', 57: 'sum butterfly', 58: 'log, logit bits', 59: 'exp-table gathers issued'})
        order = [50, 51, 52, 53] + sorted(k for k in names if k < 20) + [34] + [54, 55, 56, 57, 58, 59] + [35, 36, 37, 38]
        if clk[41] > clk[38]:
            names.update(second); order += [41, 42, 43, 44, 45, 46, 20, 21, 22, 23]
        order += [40]
        print(f'    ({ENVS} envs, rows per workgroup: {os.environ.get("BL_MLP_ROWS", "library default")})')
        for i in order:
            print(f'   {names[i]:26s} +{clk[i] - prev:7d}'); prev = clk[i]
