"""The entry point that allocates nothing (vca_colsum_add_noalloc_f32: partial sums in a fixed device array of the
library) against the entry point it is a variant of (vca_colsum_add_f32 with a caller-supplied workspace), through
the C ABI: two calls in a row, the second with a smaller input, bit for bit; and the size limit."""
import ctypes

import pytest
import torch

gpu = pytest.mark.gpu


def _both(L, P, C, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(P, C, generator=g).cuda()
    base = torch.randn(C, generator=g).cuda()
    s = torch.cuda.current_stream().cuda_stream
    want = base.clone()
    part = torch.empty((L.vca_colsum_groups(P, C), C), device='cuda', dtype=torch.float32)
    assert L.vca_colsum_add_f32(P, C, x.data_ptr(), part.data_ptr(), want.data_ptr(), s) == 0
    got = base.clone()
    torch.cuda.synchronize()
    # torch's counter only sees its own caching allocator; the driver's free-byte count (hipMemGetInfo) also sees a
    # hipMalloc / hipMallocAsync made inside the library
    before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    assert L.vca_colsum_add_noalloc_f32(P, C, x.data_ptr(), got.data_ptr(), s) == 0
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] == free_before
    return got, want, x, base


@gpu
def test_no_alloc_colsum_matches_workspace_variant():
    from vcagan_amd import _lib
    L = _lib.lib()
    assert L.vca_noalloc_init() == 0
    for P, C, seed in ((48000, 512, 1), (333, 96, 2), (7, 1, 3)):     # a large call, then smaller ones behind it
        got, want, x, base = _both(L, P, C, seed)
        assert torch.equal(got, want), (P, C, float((got - want).abs().max()))
        ref = base.double() + x.double().sum(0)
        assert float((got.double() - ref).abs().max()) <= 1e-5 * float(x.abs().sum(0).max())


@gpu
def test_no_alloc_colsum_rejects_oversize():
    from vcagan_amd import _lib
    L = _lib.lib()
    C = 8192
    P = 4096
    assert L.vca_colsum_groups(P, C) * C > L.vca_colsum_noalloc_max_floats()
    x = torch.zeros(P, C, device='cuda')
    d = torch.zeros(C, device='cuda')
    rc = L.vca_colsum_add_noalloc_f32(P, C, x.data_ptr(), d.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == -27
    torch.cuda.synchronize()
    assert float(d.abs().max()) == 0.0


@gpu
def test_no_alloc_colsum_under_graph_capture():
    """After vca_noalloc_init() the entry point can be captured into a HIP graph (where a hipMalloc would fail the
    capture) and replayed: the replays give the workspace variant's sums bit for bit, the second on new contents of
    the same input buffer."""
    from vcagan_amd import _lib
    L = _lib.lib()
    assert L.vca_noalloc_init() == 0
    P, C = 6000, 256
    g = torch.Generator(device='cpu').manual_seed(11)
    x = torch.randn(P, C, generator=g).cuda()
    dst = torch.zeros(C, device='cuda')
    part = torch.empty((L.vca_colsum_groups(P, C), C), device='cuda', dtype=torch.float32)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        rc = L.vca_colsum_add_noalloc_f32(P, C, x.data_ptr(), dst.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    for rep in range(2):
        if rep:
            x.copy_(torch.randn(P, C, generator=g))
        dst.zero_()
        graph.replay()
        torch.cuda.synchronize()
        want = torch.zeros(C, device='cuda')
        assert L.vca_colsum_add_f32(P, C, x.data_ptr(), part.data_ptr(), want.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        assert torch.equal(dst, want), rep
