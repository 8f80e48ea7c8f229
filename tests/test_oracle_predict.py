"""Known answers for the oracle's re-projection (prediction) error (oracle/sm_oracle.c:
smo_prediction_error; StcEvaluate.cpp:588-758, Warp1D.cpp:25-257), worked out by hand from the
reference's statements.  The reference ships no output of this evaluation: parity unpinned."""
import numpy as np

from oracle import oracle as O


def row(vals):
    return np.array(vals, np.uint8).reshape(1, -1, 1)


def test_forward_warp_integer_shift():
    ref = row([10, 20, 30, 40])
    fd = np.ones((1, 4), np.float32)
    rms, vis, out = O.prediction_error(ref, row([20, 30, 40, 255]), fd, 1.0)
    # every pixel moves one to the left; nothing lands on the last one: it keeps the empty colour's low byte
    assert out.ravel().tolist() == [20, 30, 40, 255] and rms == 0.0 and vis == 1.0


def test_forward_warp_half_pixel_lines_and_final_splat():
    # lines between x - 0.5 and x + 0.5 are drawn over ROUND()ed end points (overlapping), values
    # truncated after + 0.5; the last pixel has no right neighbour and is splatted at 3 - ROUND(0.5) = 2
    ref = row([10, 20, 30, 40])
    fd = np.full((1, 4), 0.5, np.float32)
    _, _, out = O.prediction_error(ref, ref, fd, 1.0)
    assert out.ravel().tolist() == [15, 25, 40, 45]
    # without line interpolation: splats at x - ROUND(0.5) = x - 1
    _, _, out = O.prediction_error(ref, ref, fd, 1.0, lin_interp=0)
    assert out.ravel().tolist() == [20, 30, 40, 255]
    # across a disparity jump >= eval_disp_gap no line is drawn
    fd = np.array([[0, 0, 2, 2]], np.float32)
    _, _, out = O.prediction_error(ref, ref, fd, 1.0)
    assert out.ravel().tolist() == [30, 40, 255, 255]          # 10, 20 are overwritten by the nearer 30, 40


def test_negative_shift_walks_right_to_left():
    ref = row([10, 20, 30, 40])
    fd = np.ones((1, 4), np.float32)
    _, _, out = O.prediction_error(ref, ref, fd, -1.0)
    assert out.ravel().tolist() == [255, 10, 20, 30]


def test_rms_and_visibility_colour():
    ref = np.zeros((1, 4, 4), np.uint8)
    ref[..., 3] = 255
    ref[0, :, 0] = [10, 20, 30, 40]
    frame = np.array(ref, copy=True)
    frame[0, :, 0] = [23, 30, 40, 0]
    fd = np.ones((1, 4), np.float32)
    rms, vis, out = O.prediction_error(ref, frame, fd, 1.0)
    # predicted B = 20, 30, 40 against 23, 30, 40; the last pixel keeps alpha 0 and is left out;
    # sum2 = 9 over 3 colour channels and 3 visible pixels
    assert out[0, :, 3].tolist() == [255, 255, 255, 0]
    assert out[0, 3].tolist() == [0xff, 0xc0, 0xff, 0x00]       # eval_empty_color 0x00ffc0ff, little endian
    assert vis == 0.75 and rms == float(np.sqrt(np.float32(9.0) / 3 / 3))
    # eval_predict_diff: the image becomes 128 + diff * scale
    _, _, out = O.prediction_error(ref, frame, fd, 1.0, predict_diff=2)
    assert out[0, :3, 0].tolist() == [128 - 6, 128, 128]


def test_inverse_warp_linear_at_borders_and_cubic_inside():
    frame = row([0, 10, 40, 90, 160, 250])                      # x^2 * 10
    ref = row([0] * 6)
    fd = np.full((1, 6), -0.5, np.float32)                      # dst(x) = frame(x + 0.5)
    _, vis, out = O.prediction_error(ref, frame, fd, 1.0, predict_type=2, disp_gap=0.0)
    # x = 0 and x = 4: linear (no four neighbours): 5, 205.  x = 1..3: CubicInterpolate as written
    # (Warp1D.cpp:130-142; its slopes are v2 - v0 and v1 - v3, not halved), f = 0.5:
    #   x = 1: v = 0, 10, 40, 90:    10 + 30*.25*2 + 40*.125 -  80*.125 = 20
    #   x = 2: v = 10, 40, 90, 160:  40 + 50*.25*2 + 80*.125 - 120*.125 = 60
    #   x = 3: v = 40, 90, 160, 250: 90 + 70*.25*2 + 120*.125 - 160*.125 = 120
    # x = 5 falls outside the frame and keeps the empty colour
    assert out.ravel().tolist() == [5, 20, 60, 120, 205, 255] and vis == 1.0


def test_partial_shuffle_absorbs_one_pixel_misalignment():
    r = np.random.RandomState(3)
    ref = r.randint(0, 256, (12, 40, 1)).astype(np.uint8)
    frame = np.roll(ref, 1, axis=1)
    fd = np.zeros((12, 40), np.float32)
    plain, _, _ = O.prediction_error(ref, frame, fd, 1.0)
    shuffled, _, _ = O.prediction_error(ref, frame, fd, 1.0, partial_shuffle=1.0)
    # with the full 3x3 [min, max] interval of both images every interval pair overlaps
    assert plain > 50 and shuffled == 0.0


def test_predict_slots():
    # tsukuba/param_in.txt: five frames, frame_ref 2, frame_match 3 (StcEvaluate.cpp:742-756)
    assert [O.predict_slot(f, 2, 3) for f in range(5)] == [None, "near", None, "match", "far"]
    assert [O.predict_slot(f, 0, 2) for f in range(5)] == [None, "middle", "match", "far", "far"]
