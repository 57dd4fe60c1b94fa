"""The host's one wait per step (`_wrapper._SentinelEvent`: block sums appearing in pinned memory) must be BOUNDED: a kernel
that never stores, a stream that is stuck, or a device fault raise a RuntimeError instead of spinning a core for good
(round-4 verdict / advisor finding; the reference's blocking `.item()` of isect_tiles.cu:200 raises on a HIP error too)."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def _pinned(n=64):
    buf = torch.empty(n, dtype=torch.int32, pin_memory=True)
    buf.fill_(-1)
    return buf


def test_never_written_buffer_raises_instead_of_hanging():
    from gscodec_studio_amd import _wrapper as W

    torch.cuda.synchronize()
    ev = W._SentinelEvent(_pinned(), what="test sums")
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="never arrived"):
        W._wait_event(ev)
    assert time.perf_counter() - t0 < 2.0


def test_busy_stream_times_out():
    from gscodec_studio_amd import _wrapper as W

    ev = W._SentinelEvent(_pinned(), what="test sums")
    torch.cuda._sleep(int(2.4e9 * 1.5))  # ~1.5 s of GPU time queued in front
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="timed out"):
        ev.synchronize(timeout_s=0.2)
    assert 0.15 < time.perf_counter() - t0 < 1.0
    torch.cuda.synchronize()


def test_late_store_is_seen():
    from gscodec_studio_amd import _wrapper as W

    buf = _pinned()
    src = torch.arange(64, dtype=torch.int32, device="cuda")
    ev = W._SentinelEvent(buf, what="test sums")
    torch.cuda._sleep(int(2.4e9 * 0.05))  # the store comes ~50 ms late: past the spin phase, into the naps
    buf.copy_(src, non_blocking=True)
    W._wait_event(ev)
    assert ev.query() and int(buf[-1]) == 63
    torch.cuda.synchronize()


def test_partial_store_is_not_taken_for_complete():
    from gscodec_studio_amd import _wrapper as W

    buf = _pinned()
    buf[0] = 5
    buf[-1] = 7
    assert not W._SentinelEvent(buf).query()
    buf[1:-1] = 0
    assert W._SentinelEvent(buf).query()
