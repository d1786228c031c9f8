"""The EXACT posterior marginals of BASELINE config 3 (Circular poses, four door landmarks, sightings with unknown
association) by forward-backward on a fine grid of the circle -- nothing sampled, nothing shared with either
implementation.  The model is the factor graph `generateCircularDoors` builds: x0 ~ N(0, 0.1) wrapped; x_i - x_{i-1} ~
N(2 pi / 50, 0.05) wrapped; at every sighting pose one of the four doors (theta_k, known to 0.01 rad) is seen at bearing
difference dz with noise 0.1, each door with probability 1/4 (multihypo = [1, .25, .25, .25, .25]).  The marginal of a
pose is the product of the forward and the backward message; a wrapped-normal transition is a circular convolution (FFT).

What it settles: the share of posterior mass within 0.35 rad of the true pose.  The doors are 1.6 rad apart (and 1.48 rad
across +-pi), so a trajectory shifted by one door spacing explains every sighting almost as well; which alias wins is
decided by the x0 prior, whose pull decays along the odometry (0.05 sqrt(i) rad of accumulated noise)."""
import numpy as np

DOORS = np.array([-2.4, -0.8, 0.8, 2.4])
STEP = 2 * np.pi / 50


def _wrapped_normal(grid, mu, sigma):
    d = (grid - mu + np.pi) % (2 * np.pi) - np.pi
    p = sum(np.exp(-0.5 * ((d + 2 * np.pi * k) / sigma) ** 2) for k in (-1, 0, 1))
    return p / p.sum()


def exact_marginals(nposes, sight_every, M=7200):
    """(grid, marginals[nposes, M]) of the pose angles"""
    grid = -np.pi + (np.arange(M) + 0.5) * (2 * np.pi / M)
    # transition x_i = x_{i-1} + step + noise: convolution kernel over the angle difference
    kern = _wrapped_normal(grid, -np.pi + 0.5 * (2 * np.pi / M) + STEP, 0.05)  # offset measured from grid[0]
    fk = np.fft.rfft(kern)
    conv = lambda p: np.fft.irfft(np.fft.rfft(p) * fk, M)                       # forward: p(x_i) from p(x_{i-1})
    corr = lambda p: np.fft.irfft(np.fft.rfft(p) * np.conj(fk), M)              # backward
    emis = np.ones((nposes, M))
    for i in range(0, nposes, sight_every):
        xi = (i * STEP + np.pi) % (2 * np.pi) - np.pi
        dz = min(((d - xi + np.pi) % (2 * np.pi) - np.pi for d in DOORS), key=abs)
        sig = np.hypot(0.1, 0.01)
        e = sum(0.25 * _wrapped_normal(grid, th - dz, sig) for th in DOORS)     # x + dz = theta_k + noise
        emis[i] = e / e.max()
    fwd = np.empty((nposes, M))
    a = _wrapped_normal(grid, 0.0, 0.1) * emis[0]
    fwd[0] = a / a.sum()
    for i in range(1, nposes):
        a = np.maximum(conv(fwd[i - 1]), 0) * emis[i]
        fwd[i] = a / a.sum()
    bwd = np.ones(M) / M
    out = np.empty((nposes, M))
    out[-1] = fwd[-1]
    for i in range(nposes - 2, -1, -1):
        b = np.maximum(corr(bwd * emis[i + 1]), 0)
        bwd = b / b.sum()
        m = fwd[i] * bwd
        out[i] = m / m.sum()
    return grid, out


def exact_share_at_truth(nposes, sight_every, halfwidth=0.35, M=7200):
    """per pose: exact posterior mass within `halfwidth` of the true angle i * 2 pi / 50"""
    grid, marg = exact_marginals(nposes, sight_every, M)
    truth = (np.arange(nposes) * STEP + np.pi) % (2 * np.pi) - np.pi
    d = np.abs((grid[None, :] - truth[:, None] + np.pi) % (2 * np.pi) - np.pi)
    return (marg * (d < halfwidth)).sum(axis=1)
