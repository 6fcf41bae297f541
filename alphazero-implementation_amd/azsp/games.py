"""Host-side Game plugin surface (reference: src/alphazero_simple/base_game.py:6-62,
connect4_game.py:6-101, connect2_game.py:6-80).  Same class and method names and the same
return conventions; the rules are written fresh (vectorised numpy) and each shipped game names
the device trait (csrc/azsp_games.h) the engine runs it with."""
from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np


class BaseGame(ABC):
    #: name of the device-side trait; games without one cannot run on the engine
    device_trait: str | None = None

    @abstractmethod
    def get_init_board(self) -> np.ndarray: ...

    @abstractmethod
    def get_board_size(self): ...

    @abstractmethod
    def get_action_size(self) -> int: ...

    @abstractmethod
    def get_next_state(self, board: np.ndarray, player: int, action: int) -> tuple[np.ndarray, int]: ...

    @abstractmethod
    def has_legal_moves(self, board: np.ndarray) -> bool: ...

    @abstractmethod
    def get_valid_moves(self, board: np.ndarray) -> list[int]: ...

    @abstractmethod
    def is_win(self, board: np.ndarray, player: int) -> bool: ...

    @abstractmethod
    def get_reward_for_player(self, board: np.ndarray, player: int): ...

    def get_canonical_board(self, board: np.ndarray, player: int) -> np.ndarray:
        return player * board

    def _reward(self, board, player):
        if self.is_win(board, player):
            return 1
        if self.is_win(board, -player):
            return -1
        if self.has_legal_moves(board):
            return None
        return 0


def _window_table(rows: int, cols: int, k: int) -> np.ndarray:
    idx = np.arange(rows * cols).reshape(rows, cols)
    wins = []
    for dr, dc in ((0, 1), (1, 0), (1, 1), (-1, 1)):
        for r in range(rows):
            for c in range(cols):
                r2, c2 = r + dr * (k - 1), c + dc * (k - 1)
                if 0 <= r2 < rows and 0 <= c2 < cols:
                    wins.append([idx[r + i * dr, c + i * dc] for i in range(k)])
    return np.array(wins, dtype=np.int64)


class Connect4Game(BaseGame):
    """6x7 Connect-4: row 0 is the top, pieces fall to the largest empty row index."""
    device_trait = "connect4"

    def __init__(self):
        self.rows, self.columns, self.win_length = 6, 7, 4
        self._windows = _window_table(6, 7, 4)

    def get_init_board(self):
        return np.zeros((self.rows, self.columns), dtype=int)

    def get_board_size(self):
        return (self.rows, self.columns)

    def get_action_size(self):
        return self.columns

    def get_next_state(self, board, player, action):
        b = np.copy(board)
        empty = np.flatnonzero(b[:, action] == 0)
        if empty.size:                      # a full column leaves the board unchanged (no legality check)
            b[empty[-1], action] = player
        return b, -player

    def has_legal_moves(self, board):
        return bool((np.asarray(board)[0] == 0).any())

    def get_valid_moves(self, board):
        return [int(v) for v in (np.asarray(board)[0] == 0)]

    def is_win(self, board, player):
        cells = np.asarray(board).reshape(-1)[self._windows]
        return bool((cells == player).all(axis=1).any())

    def get_reward_for_player(self, board, player):
        return self._reward(board, player)


class Connect2Game(BaseGame):
    """1x4 strip, two in a row wins (the reference's toy game for MCTS tests)."""
    device_trait = "connect2"

    def __init__(self):
        self.columns, self.win = 4, 2

    def get_init_board(self):
        return np.zeros((self.columns,), dtype=int)

    def get_board_size(self):
        return self.columns

    def get_action_size(self):
        return self.columns

    def get_next_state(self, board, player, action):
        b = np.copy(board)
        b[action] = player
        return b, -player

    def has_legal_moves(self, board):
        return bool((np.asarray(board) == 0).any())

    def get_valid_moves(self, board):
        return [int(v) for v in (np.asarray(board) == 0)]

    def is_win(self, board, player):
        b = np.asarray(board) == player
        return bool((b[:-1] & b[1:]).any())

    def get_reward_for_player(self, board, player):
        return self._reward(board, player)


class GomokuGame(BaseGame):
    """9x9 Gomoku, five (or more) in a row wins -- BASELINE.json configs[4].  The reference ships no such game:
    this is a NEW plugin written against its BaseGame contract (base_game.py:6-62), with the conventions of its
    Connect4Game: actions are cell indices row*9 + col, `get_next_state` does no legality check (an occupied
    cell leaves the board unchanged), a win is tested before a draw."""
    device_trait = "gomoku9"

    def __init__(self, size: int = 9, win_length: int = 5):
        if (size, win_length) != (9, 5):
            raise NotImplementedError("the device trait is 9x9 / five in a row")
        self.rows = self.columns = size
        self.win_length = win_length
        self._windows = _window_table(size, size, win_length)

    def get_init_board(self):
        return np.zeros((self.rows, self.columns), dtype=int)

    def get_board_size(self):
        return (self.rows, self.columns)

    def get_action_size(self):
        return self.rows * self.columns

    def get_next_state(self, board, player, action):
        b = np.copy(board)
        r, c = divmod(int(action), self.columns)
        if b[r, c] == 0:
            b[r, c] = player
        return b, -player

    def has_legal_moves(self, board):
        return bool((np.asarray(board) == 0).any())

    def get_valid_moves(self, board):
        return [int(v) for v in (np.asarray(board).reshape(-1) == 0)]

    def is_win(self, board, player):
        cells = np.asarray(board).reshape(-1)[self._windows]
        return bool((cells == player).all(axis=1).any())

    def get_reward_for_player(self, board, player):
        return self._reward(board, player)
