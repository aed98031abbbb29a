"""boardlaw_amd: MI355X-native vectorised-MCTS self-play for boardlaw's Hex.

    from boardlaw_amd import hex, mcts, networks
    worlds = hex.Hex.initial(4096, 9)                       # device='cuda'
    agent = mcts.MCTSAgent(networks.FCModel(worlds.obs_space, worlds.action_space, 512, 4).cuda(), n_nodes=64)
    decisions = agent(worlds); worlds, transitions = worlds.step(decisions.actions)

The kernels live in libboardlaw_amd.so (boardlaw_amd/csrc, C ABI in include/boardlaw_amd.h); build it with
`python -m boardlaw_amd.build`.  There is no CPU or PyTorch fallback for them."""
__version__ = '0.1.0'
