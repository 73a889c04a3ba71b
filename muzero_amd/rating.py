"""Elo ratings with the reference's semantics (rating.py:18-69): expected score on the 400-point logistic scale and the
K-factor update of both players after one game."""

_ELO_C = 1.0 / 400.0


def estimate_win_probability(ra, rb, c_elo: float = 1 / 400) -> float:
    """Expected score of player A against player B."""
    return 1.0 / (1 + 10 ** ((rb - ra) * c_elo))


def compute_elo_rating(winner, ra=0, rb=0, k=32):
    """Ratings (A, B) after a game won by A (`winner == 0`) or B (`winner == 1`); `None` leaves them unchanged.
    Each rating moves by k * (actual score - expected score)."""
    if winner is None:
        return (ra, rb)
    if not isinstance(winner, int) or winner not in (0, 1):
        raise ValueError(f'Expect input argument `winner` to be [0, 1], got {winner}')
    expected = (estimate_win_probability(ra, rb, _ELO_C), estimate_win_probability(rb, ra, _ELO_C))
    actual = (1 - winner, winner)
    return tuple(r + k * (s - e) for r, s, e in zip((ra, rb), actual, expected))
