"""The oracle's restatement of Histogram1D as ComputeMatchQuality draws it (Histogram1D.cpp:32-139,
StcEvaluate.cpp:848-861) against answers worked out by hand from the reference's statements.  Parity unpinned:
the reference holds no such figure."""
import numpy as np


def test_counts_clamp_scale_and_figure(oracle):
    img = np.array([[0.5, 1.5, 1.6, 3.9, 7.0, -1.0]], np.float32)      # bins of width 1 over [0, 4): 2, 2, 0, 2
    fig, vs = oracle.cost_histogram(img, None, 0.0, 0.0, 4.0, 4, 4)    # (values beyond the range go to the end bins)
    assert vs == 2.0                                                   # height / tallest bin (:118-119)
    # bars of min(height - 1, int(count * scale + 0.5)) = 3, 3, 0, 3 black rows from the bottom, white above (:124-138)
    want = np.array([[255, 255, 255, 255], [0, 0, 255, 0], [0, 0, 255, 0], [0, 0, 255, 0]], np.uint8)
    assert np.array_equal(fig, want)


def test_mask_zero_ends_the_row_and_scale_is_reused(oracle):
    img = np.array([[0.5, 1.5, 2.5, 3.5], [0.5, 0.6, 0.7, 3.5]], np.float32)
    mask = np.array([[255, 0, 255, 255], [255, 255, 255, 0]], np.uint8)
    # row 0 stops at its second pixel (Histogram1D.cpp:86-87 leaves the loop), row 1 at its last: counts 4, 0, 0, 0
    fig, vs = oracle.cost_histogram(img, mask, 0.5, 0.0, 4.0, 4, 8)    # scale given: bars of int(4 * 0.5 + 0.5) = 2
    assert vs == 0.5
    assert [int((fig[:, b] == 0).sum()) for b in range(4)] == [2, 0, 0, 0]
    assert np.array_equal(fig[-2:, 0], [0, 0]) and (fig[:-2] == 255).all()
