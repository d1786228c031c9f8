"""CPU: the exact posterior of config 3 (tests/doors_exact.py) -- sanity of the forward-backward itself and what it says
about the aliasing of the doors (the GPU leg compares the solver with it at 200 and 2000 poses: test_gpu_doors_exact.py)."""
import numpy as np

from doors_exact import exact_marginals, exact_share_at_truth


def test_marginals_are_distributions_and_the_prior_pose_is_the_prior():
    grid, m = exact_marginals(6, 3, M=3600)
    assert np.allclose(m.sum(axis=1), 1.0) and (m >= 0).all()
    # x0: prior N(0, 0.1) times a sighting that has a door at the truth: the mass stays at 0
    assert abs(grid[m[0].argmax()]) < 0.05
    assert exact_share_at_truth(6, 3, M=3600).min() > 0.95


def test_short_chains_are_resolved_long_ones_alias():
    assert exact_share_at_truth(26, 25, M=3600).min() > 0.95
    s200, s2000 = exact_share_at_truth(200, 25, M=3600), exact_share_at_truth(2000, 25, M=3600)
    # Every sighting is consistent with the trajectory shifted by a door spacing (1.6 rad) -- but the chain is rigid between
    # sightings (25 steps of 0.05 rad of noise against 1.6 rad to the next alias) and x0 is pinned by its prior, so the
    # EXACT posterior stays on the true alias all along the chain: the aliases do not explain a low share at the truth.
    assert np.median(s200) > 0.95 and s200.min() > 0.8
    assert np.median(s2000) > 0.95 and s2000.min() > 0.8
    print(f"exact share at the truth: 200 poses median {np.median(s200):.3f} min {s200.min():.3f}; 2000 poses median {np.median(s2000):.3f} min {s2000.min():.3f}")
