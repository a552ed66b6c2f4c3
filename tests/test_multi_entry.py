"""The C-side multi-device entry (lanczos_multi.hip): partition arithmetic without a GPU; on the 1-GPU box the scheduler
itself with two contexts on device 0 (host path) and the root path with one device (no exchange partner).  The 8-GPU
exchange over RCCL cannot run on this pool's boxes: its code path is exercised as far as one device allows."""
import numpy as np
import pytest

import lanczos_hls_amd as L
import oracle_lib as O
import patterns as P


def test_partition_frames_matches_block_partition():
    import lanczos_hls_amd.sharding as sh
    for n in (0, 1, 5, 8, 64, 65):
        for w in (1, 2, 3, 8):
            assert [L.partition_frames(n, w, i) for i in range(w)] == sh.frame_shards(n, w)
    assert [L.partition_frames(64, 8, i) for i in range(8)] == [(8 * i, 8) for i in range(8)]      # BASELINE config 4
    with pytest.raises(L.LanczosError):
        L.partition_frames(4, 2, 2)


def test_partition_rows_config5_and_tiny_frames():
    d = L.make_desc(3840, 2160, 4, 2, 1, 4, bytes_per_sample=2)                                    # BASELINE config 5
    parts = [L.partition_rows(d, 8, i) for i in range(8)]
    assert [p[0] for p in parts] == [540 * i for i in range(8)] and all(p[1] == 540 for p in parts)
    for (r0, rows, i0, n) in parts:
        assert (i0, n) == L.strip_input_rows(d, r0, rows)
        assert i0 == max(0, r0 // 2 - 3) and i0 + n - 1 == min(2159, (r0 + rows - 1) // 2 + 4)
    assert sum(p[3] for p in parts) <= 2160 * 1.04                                                  # ~3 % halo
    # tiny frame, many parts: the first strip keeps the in-place prefix rows, later strips may be empty, rows are covered once
    d2 = L.make_desc(32, 10, 3, 2, 1, 3)
    p2 = [L.partition_rows(d2, 8, i) for i in range(8)]
    assert p2[0][1] >= L.inplace_rows(d2) and sum(p[1] for p in p2) == d2.out_h
    pos = 0
    for (r0, rows, _i0, _n) in p2:
        assert r0 == pos or rows == 0
        pos += rows
    # HLS mode has no prefix: plain equal strips
    d3 = L.make_desc(64, 40, 3, 2, 1, 3, 1, L.MODE_HLS)
    assert [L.partition_rows(d3, 4, i)[:2] for i in range(4)] == [(20 * i, 20) for i in range(4)]


@pytest.mark.gpu
def test_multi_host_path_two_contexts_on_one_device():
    frames = np.stack([P.noise(72, 96, 3, seed=500 + i) for i in range(5)])
    want = [O.expected_hwc_u8(O.cfg(96, 72, 192, 144, 3, 3, 2, 1), f) for f in frames]
    m = L.MultiContext([0, 0])
    try:
        for split in (L.SPLIT_FRAMES, L.SPLIT_ROWS):
            got = m.resample(frames, 2, 1, 3, L.MODE_EXACT, split)
            for i in range(5):
                assert np.array_equal(got[i], want[i]), (split, i)
        one = m.resample(frames[:1], 2, 1, 3, L.MODE_EXACT, L.SPLIT_FRAMES)        # fewer frames than devices
        assert np.array_equal(one[0], want[0])
        u16 = np.stack([P.noise(48, 64, 4, seed=9, dtype=np.uint16)] * 2)
        got = m.resample(u16, 2, 1, 4, L.MODE_EXACT, L.SPLIT_ROWS)
        assert np.array_equal(got[0], O.expected_hwc_u16(O.cfg(64, 48, 128, 96, 4, 4, 2, 1), u16[0]))
    finally:
        m.close()


@pytest.mark.gpu
def test_multi_root_path_single_device():
    import torch
    frames = np.stack([P.gradient_noise(64, 80, 3, seed=70 + i) for i in range(3)])
    d = L.make_desc(80, 64, 3, 2, 1, 3, 1, L.MODE_EXACT)
    x = torch.from_numpy(frames).cuda()
    y = torch.zeros((3, 128, 160, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    m = L.MultiContext([0])
    try:
        for split in (L.SPLIT_FRAMES, L.SPLIT_ROWS):
            y.zero_()
            torch.cuda.synchronize()
            cm, tm = m.resample_root(d, x.data_ptr(), y.data_ptr(), 3, split)
            assert 0 <= cm <= tm
            got = y.cpu().numpy()
            for i in range(3):
                assert np.array_equal(got[i], O.expected_hwc_u8(O.cfg(80, 64, 160, 128, 3, 3, 2, 1), frames[i]))
    finally:
        m.close()
    with pytest.raises(L.LanczosError):
        L.MultiContext([0, 99])                                                     # no such device
