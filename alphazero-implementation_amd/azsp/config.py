"""Host-side mirror of the reference configuration object
(reference: src/alphazero_simple/config.py:4-18; same field names and meaning)."""
from dataclasses import dataclass


@dataclass
class AlphaZeroConfig:
    batch_size: int
    num_iterations: int
    num_simulations: int          # MCTS simulations per move (read on the hot path)
    num_episodes: int             # concurrent self-play games = episodes per call (hot path)
    num_iters_for_train_history: int
    epochs: int
    background_generation: bool = False
