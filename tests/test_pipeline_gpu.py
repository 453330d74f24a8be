"""SURVEY f3 on the GPU: device-side source_map sort + run-wise pointer scatter (K11 sorted form) and the upload-ahead loader.
Reference semantics: common/Utils.py:344-355 (one-hot build_map) followed by CaSE/Model.py:43 (bmm) == a scatter-add over ids."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ids(B, S, V, seed, oov=True):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, V, (B, S), generator=g)
    ids[:, : S // 3] = ids[:, :1]  # a long run of one token (crosses the 256-key chunks when S is large)
    if oov and S > 4:
        ids[0, 1], ids[-1, S - 1] = V + 5, -3  # ids outside the vocabulary are dropped, as in case_copy_scatter_fwd
    return ids


@pytest.mark.parametrize("B,S,V", [(3, 1, 50), (2, 300, 97), (4, 3904, 30522), (2, 20544, 30522), (1, 32768, 131071)])
def test_source_sort_matches_a_host_sort(B, S, V):
    from case_rg_amd import ops
    ids = _ids(B, S, V, 1)
    srt = ops.SortedSource(ids.cuda(), V)
    got = srt.keys.cpu().numpy().view(np.uint32)
    tok = ids.numpy()
    want = np.where((tok >= 0) & (tok < V), (tok.astype(np.int64) << 15) | np.arange(S)[None, :], 0xFFFFFFFF).astype(np.uint32)
    assert np.array_equal(got, np.sort(want, axis=1))


@pytest.mark.parametrize("B,T,S,V", [(2, 3, 300, 97), (4, 5, 3904, 30522), (2, 1, 20544, 30522)])
def test_sorted_scatter_equals_index_add_and_is_reproducible(B, T, S, V):
    from case_rg_amd import ops
    ids = _ids(B, S, V, 2).cuda()
    g = torch.Generator().manual_seed(3)
    w = torch.rand(B, T, S, generator=g).cuda()
    w[:, :, ::7] = 0.0
    base = torch.rand(B, T, V, generator=g).cuda()
    srt = ops.SortedSource(ids, V)
    w1 = w.clone().requires_grad_(True)
    d_sorted = ops.copy_scatter(srt, w1, V, base)
    d_atomic = ops.copy_scatter(ids, w, V, base)
    ok = (ids >= 0) & (ids < V)
    want = base.double().clone()
    want.scatter_add_(2, ids.clamp(0, V - 1).unsqueeze(1).expand(B, T, S), (w * ok.unsqueeze(1)).double())
    assert torch.allclose(d_sorted.double(), want, rtol=2e-6, atol=1e-6)  # chunk-wise partial sums: tighter than the atomic form
    assert torch.allclose(d_atomic.double(), want, rtol=2e-4, atol=1e-6)  # one f32 atomic per element: ~7 k serial adds into the long run's token
    again = ops.copy_scatter(srt, w, V, base)
    assert torch.equal(d_sorted, again), "run-wise adds in a fixed order: bit-reproducible"
    gout = torch.rand(B, T, V, generator=g).cuda()
    d_sorted.backward(gout)
    want_g = torch.gather(gout, 2, ids.clamp(0, V - 1).unsqueeze(1).expand(B, T, S)) * ok.unsqueeze(1)
    assert torch.equal(w1.grad, want_g)
    with pytest.raises(ValueError):
        ops.copy_scatter(srt, w, V + 1)


def test_sort_rejects_what_does_not_fit_the_key():
    from case_rg_amd import ops
    assert not ops.SortedSource.fits(torch.zeros(2, 40000, dtype=torch.int64), 100)
    assert not ops.SortedSource.fits(torch.zeros(2, 10, dtype=torch.int64), 200000)
    with pytest.raises(RuntimeError, match="case_source_sort"):
        ops.SortedSource(torch.zeros(1, 40000, dtype=torch.int64).cuda(), 100)


def test_prefetcher_uploads_every_batch_in_order():
    from case_rg_amd.utils.pipeline import DevicePrefetcher
    batches = [{"id": torch.arange(4) + 4 * i, "x": torch.full((4, 1000), float(i)), "tag": "b%d" % i} for i in range(5)]
    seen = []
    for b in DevicePrefetcher(batches):
        assert b["id"].is_cuda and b["x"].is_cuda and isinstance(b["tag"], str)
        seen.append((b["id"].cpu().tolist(), float(b["x"].sum().item()), b["tag"]))
    assert seen == [((torch.arange(4) + 4 * i).tolist(), 4000.0 * i, "b%d" % i) for i in range(5)]
    assert list(DevicePrefetcher([])) == []
