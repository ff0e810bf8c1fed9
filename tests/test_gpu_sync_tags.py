"""-m gpu: pps tags -- the GNU Radio source block's per-sample meta loop (caribouLiteSource_impl.cc:113-119) as an
ordered device compaction (clhip_sync_tags), bit for bit against the oracle's loop: every plane length around the
kernel's 16-byte groups, 16 KiB rounds and 64 KiB tiles, every base alignment, meta values other than 0 / 1
(slots a re-sync left untouched hold what they held: caribou_smi.c:382-389), a capacity smaller than the count."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def H():
    import torch
    from cariboulite_amd import hip
    assert torch.cuda.is_available() and hip.require_gpu().startswith("gfx950")
    return hip


def _run(H, meta, cap=None, misalign=0):
    """tags of `meta` through the C ABI; the plane sits `misalign` bytes behind a 256-byte aligned address, between
    guard zones of ones that the kernel must neither read as part of the plane nor, for the outputs, write."""
    import torch
    n = meta.size
    cap = n if cap is None else cap
    plane = torch.ones(256 + n + 256, dtype=torch.uint8, device=DEV)           # guards = 1: a stray read would tag
    if n:
        plane[misalign:misalign + n] = torch.from_numpy(meta.copy()).to(DEV)
    idx = torch.full((cap + 8,), -559038737, dtype=torch.int32, device=DEV)
    cnt = torch.full((3,), -7, dtype=torch.int32, device=DEV)
    ws_bytes = H.lib().clhip_sync_tags_ws_bytes(n)
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=DEV)
    H.sync_tags(plane.data_ptr() + misalign, n, idx, cap, cnt[1:], ws if ws_bytes else None)
    torch.cuda.synchronize()
    c = cnt.cpu().numpy()
    assert c[0] == -7 and c[2] == -7
    got = idx.cpu().numpy()
    k = int(c[1])
    assert (got[min(k, cap):] == -559038737).all(), "wrote past min(count, cap)"
    return got[:min(k, cap)].view(np.uint32), k


LENGTHS = [0, 1, 15, 16, 17, 255, 4095, 4096, 4097, 16383, 16384, 16385, 65535, 65536, 65537, 131072, 131080,
           262144 - 15, 262144 - 14, 262144, 262145, 3 * 65536 + 5, 1000003]


@pytest.mark.parametrize("n", LENGTHS)
def test_tags_equal_the_work_loop(H, orc, n):
    rng = np.random.default_rng(1000 + n)
    for density, mis in ((0.0, 0), (1e-4, 3), (0.02, 15), (0.5, 8), (1.0, 1)):
        r = rng.random(n)
        meta = np.where(r < density, 1, rng.choice(np.array([0, 0, 0, 2, 3, 0x81, 0xAA, 0xFF], np.uint8), n)).astype(np.uint8)
        want, k = orc.sync_tags(meta)
        got, kg = _run(H, meta, misalign=mis)
        assert kg == k, (n, density, mis)
        assert np.array_equal(got, want), (n, density, mis)


def test_every_base_alignment(H, orc):
    rng = np.random.default_rng(5)
    for n in (100, 70000, 262144 + 77):
        meta = (rng.random(n) < 0.03).astype(np.uint8)
        meta[[0, -1]] = 1
        want, k = orc.sync_tags(meta)
        for mis in range(16):
            got, kg = _run(H, meta, misalign=mis)
            assert kg == k and np.array_equal(got, want), (n, mis)


@pytest.mark.parametrize("n", [5000, 131072, 500000])
def test_capacity_smaller_than_the_count(H, orc, n):
    rng = np.random.default_rng(n)
    meta = (rng.random(n) < 0.1).astype(np.uint8)
    for cap in (0, 1, 7, 300):
        want, k = orc.sync_tags(meta, cap)
        got, kg = _run(H, meta, cap=cap, misalign=4)
        assert kg == k and k > cap
        assert np.array_equal(got, want)


def test_the_unpack_kernels_meta_plane_with_one_pps(H, orc):
    """the stream as the FPGA sends it: the sync bit of the word of every 4 000 000th sample -> unpack -> tags."""
    import torch
    from cariboulite_amd import synth
    from gpu_util import dev_bytes
    n = 131072
    w = synth.torch_smi_words(n, torch.device(DEV), channel=0, stream=0).cpu().numpy().view(np.uint32).copy()
    w &= ~np.uint32(1)
    pps = np.array([0, 4097, 65535, 65536, 131071])
    w[pps] |= 1
    d = dev_bytes(w.view(np.uint8))
    offs = torch.zeros(1, dtype=torch.int32, device=DEV)
    iq = torch.empty((n + 2, 2), dtype=torch.int16, device=DEV)
    meta = torch.full((n + 2,), 0xAA, dtype=torch.uint8, device=DEV)
    H.smi_find_offsets(d, 4 * n, 4 * n + 4, 4 * n, 1, offs)
    H.smi_unpack(0, d, 4 * n, 4 * n + 4, 4 * n, 1, offs, H.FORMAT_CS16, iq, meta)
    idx = torch.zeros(16, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    H.sync_tags(meta, n, idx, 16, cnt)
    torch.cuda.synchronize()
    assert int(cnt.item()) == pps.size and idx.cpu().numpy()[:pps.size].tolist() == pps.tolist()
    want, k = orc.sync_tags(meta.cpu().numpy()[:n])
    assert k == pps.size and want.tolist() == pps.tolist()
