"""Elo ratings (rating.py:18-69)."""


def estimate_win_probability(ra, rb, c_elo: float = 1 / 400) -> float:
    """Estimated probability of winning from player A's perspective (rating.py:18-30)."""
    return 1.0 / (1 + 10 ** ((rb - ra) * c_elo))


def compute_elo_rating(winner, ra=0, rb=0, k=32):
    """New (elo_A, elo_B) after a game; winner 0 = player A, 1 = player B, None = unchanged (rating.py:33-69)."""
    if winner is None:
        return (ra, rb)
    if not isinstance(winner, int) or winner not in [0, 1]:
        raise ValueError(f'Expect input argument `winner` to be [0, 1], got {winner}')
    c_elo = 1.0 / 400.0
    prob_a = estimate_win_probability(ra, rb, c_elo)
    prob_b = estimate_win_probability(rb, ra, c_elo)
    if winner == 0:
        return (ra + k * (1 - prob_a), rb + k * (0 - prob_b))
    return (ra + k * (0 - prob_a), rb + k * (1 - prob_b))
