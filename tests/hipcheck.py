"""Shared comparison helpers for the -m gpu parity tests (HIP path through the C-ABI vs the oracle)."""
import numpy as np


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def compare_pyramids(g, o):
    """g: capi.CSIFT3D after run_stages(>=1); o: oracle Extractor after run(>=2). Bit-exact, every level."""
    assert g.num_octaves == o.num_octaves
    for oc in range(g.num_octaves):
        for i in range(6):
            a, b = g.gss(oc, i), o.gss(oc, i)
            assert np.array_equal(bits(a), bits(b)), ("gss", oc, i, int((bits(a) != bits(b)).sum()))
            assert g.level_info(0, oc * 6 + i) == o.level_info(0, oc * 6 + i)
        for i in range(5):
            a, b = g.dog(oc, i), o.dog(oc, i)
            assert np.array_equal(bits(a), bits(b)), ("dog", oc, i, int((bits(a) != bits(b)).sum()))


def extrema_table(e):
    return np.stack([e["octave"], e["level"], e["x"].astype(np.int32), e["y"].astype(np.int32), e["z"].astype(np.int32)], 1)


def compare_keypoints(kp, desc, okp, odesc, rms_tol=1e-4):
    """Same count, same (octave, level, x, y, z, scale, rx, ry, rz) in the same order; orientation
    frames and descriptors within fp32 reduction-order tolerance.  Returns the descriptor RMS."""
    assert len(kp) == len(okp), (len(kp), len(okp))
    for f in ("x", "y", "z", "octave", "level", "scale", "rx", "ry", "rz"):
        assert np.array_equal(kp[f], okp[f]), f
    if len(kp) == 0:
        return 0.0
    # structure tensor / mean gradient: fp32 sums in a different order
    scale = np.abs(okp["str_tensor"]).max(1, keepdims=True)
    assert (np.abs(kp["str_tensor"] - okp["str_tensor"]) <= 2e-5 * scale).all()
    assert np.allclose(kp["win"], okp["win"], rtol=0, atol=2e-5 * np.abs(okp["win"]).max())
    assert np.allclose(kp["eigvalue"], okp["eigvalue"], rtol=2e-5, atol=1e-7)
    assert np.abs(kp["Rotation"] - okp["Rotation"]).max() <= 2e-4
    rms = float(np.sqrt(np.mean((desc.astype(np.float64) - odesc.astype(np.float64)) ** 2)))
    assert rms <= rms_tol, rms
    assert np.abs(desc - odesc).max() <= 50 * rms_tol
    return rms
