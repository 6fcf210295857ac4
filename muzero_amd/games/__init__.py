"""Host-side game classes with the reference's interface (`games/game.py`, `games/tictactoe.py`,
`games/cartpole.py`) for callers that step ONE environment from Python, and the trajectory container
`GameHistory`.  Batched self-play steps environments on the device instead (muzero_amd/self_play.py)."""
from .game import Game, BoardGame, GameHistory  # noqa: F401
from .tictactoe import TicTacToe  # noqa: F401
from .cartpole import CartPole  # noqa: F401
