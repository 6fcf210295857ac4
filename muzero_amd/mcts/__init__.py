"""Drop-in for the reference's `mcts` package: put the directory that contains this package
(`muzero_amd/`) ahead of the reference on `sys.path` and `from mcts.mcts import MCTS` resolves
here; `self_play.py` / `reanalyse.py` then search through the HIP engine unchanged."""
