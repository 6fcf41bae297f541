"""azsp -- MI355X-native self-play episode generation behind the plugin surface of
pierreveron/alphazero-implementation (`BaseGame` / `BaseModel` / `MCTS` / `EpisodeGenerator`)."""
from .config import AlphaZeroConfig
from .episode import Episode, Sample
from .games import BaseGame, Connect2Game, Connect4Game, GomokuGame

__all__ = ["AlphaZeroConfig", "Episode", "Sample", "BaseGame", "Connect2Game", "Connect4Game", "GomokuGame",
           "BaseModel", "ResNet", "MCTS", "Node", "EpisodeGenerator", "Engine", "generate_dataset",
           "DeviceReplayBuffer", "save_episodes", "load_episodes", "ShardedEpisodeGenerator",
           "DistributedEpisodeGenerator", "gen1", "connect_state"]


def __getattr__(name):          # torch / libazsp are only imported when their classes are used
    if name in ("BaseModel", "ResNet", "ResBlock"):
        from . import model
        return getattr(model, name)
    if name in ("MCTS", "Node"):
        from . import mcts
        return getattr(mcts, name)
    if name == "EpisodeGenerator":
        from .episode_generator import EpisodeGenerator
        return EpisodeGenerator
    if name == "Engine":
        from .engine import Engine
        return Engine
    if name in ("ShardedEpisodeGenerator", "DistributedEpisodeGenerator"):
        from . import sharded
        return getattr(sharded, name)
    if name in ("generate_dataset", "DeviceReplayBuffer", "save_episodes", "load_episodes"):
        from . import replay
        return getattr(replay, name)
    if name in ("gen1", "connect_state"):     # first-generation surface (src/alphazero_implementation)
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
