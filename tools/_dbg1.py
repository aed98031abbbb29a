import torch, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from boardlaw_amd import networks
from boardlaw_amd.hex import Hex
from boardlaw_amd.mcts import MCTSAgent, MoveRng
w = Hex.initial(2048, 9)
net = networks.Inference(networks.FCModel(w.obs_space, w.action_space, 512, 4).cuda(), fused=True)
a = MCTSAgent(net, graph=True, n_nodes=64, rng=MoveRng())
r0, a0 = torch.cuda.memory_reserved(), torch.cuda.memory_allocated()
a(w[:1000])
print('reserved delta', torch.cuda.memory_reserved() - r0, 'allocated delta', torch.cuda.memory_allocated() - a0, 'nbytes', [g.nbytes for g in a._graphs.values()])
st = torch.cuda.memory_stats()
print({k: v for k, v in st.items() if 'reserved_bytes' in k and 'current' in k})
import subprocess
r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_reference_fixtures.py', '-q', '-x', '-k', 'learner_step_on_gpu'], capture_output=True, text=True)
print(r.stdout[-3000:])
