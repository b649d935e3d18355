"""GPU test of ilqr_solve_batched_static, the entry point that allocates nothing: its results against
ilqr_solve_batched's on two calls in a row, the second with a smaller input (the library's workspace then
still holds the first call's data).  Bar: torch.equal on every output."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ilqr_amd import _hip
    _hip.lib()
    return _hip


def _batch(B, T, seed):
    r = np.random.RandomState(seed)
    st = np.stack([r.uniform(-1, 1, B), r.uniform(-1, 1, B), r.uniform(-np.pi, np.pi, B), r.uniform(-1, 1, B)], 1)
    x0 = np.stack([st[:, 0], st[:, 1], np.sin(st[:, 2]), np.cos(st[:, 2]), st[:, 3]], 1)
    return x0, r.uniform(-1, 1, (B, T, 1))


def _model(hip, T):
    Q = 100.0 * np.eye(5)
    Q[1, 1] = Q[4, 4] = 10.0
    return hip.Model(T, Q, [[5.0]], 100.0 * np.eye(5), [0.0, 0.0, 0.0, 1.0, 0.0], None, [-5.0], [5.0])


def _outputs(B, T, dev):
    f64 = torch.float64
    return dict(xs=torch.empty((B, T + 1, 5), dtype=f64, device=dev), us=torch.empty((B, T, 1), dtype=f64, device=dev),
                J=torch.empty(B, dtype=f64, device=dev), mu=torch.empty(B, dtype=f64, device=dev),
                iters=torch.empty(B, dtype=torch.int32, device=dev),
                status=torch.empty(B, dtype=torch.int32, device=dev))


def test_static_workspace_solve_equals_the_workspace_solve_twice_in_a_row(hip):
    dev = torch.device("cuda:0")
    T = 40
    M = _model(hip, T)
    O = hip.Opts(max_iter=30)
    for B, seed in ((1500, 3), (333, 4)):  # the second call is smaller: stale data stays behind it
        assert hip.solve_workspace_bytes(M, B) <= hip.STATIC_WORKSPACE_BYTES
        x0_h, us_h = _batch(B, T, seed)
        x0, us0 = torch.tensor(x0_h, device=dev), torch.tensor(us_h, device=dev)
        ref = hip.solve(M, O, x0, us0)
        delta = torch.full((B,), 2.0, dtype=torch.float64, device=dev)
        k = torch.zeros((B, T, 1), dtype=torch.float64, device=dev)
        K = torch.zeros((B, T, 1, 5), dtype=torch.float64, device=dev)
        out = hip.solve_static(M, O, x0, us0, delta, k, K, _outputs(B, T, dev))
        torch.cuda.synchronize()
        for key in ("xs", "us", "J", "mu", "iters", "status"):
            assert torch.equal(out[key], ref[key]), (B, key)
        assert torch.equal(delta, ref["delta"]) and torch.equal(k, ref["k"]) and torch.equal(K, ref["K"])
        it = ref["iters"].cpu().numpy()
        assert it.min() < it.max()  # problems end at different iterations, as in a real batch


def test_static_workspace_solve_refuses_what_does_not_fit(hip):
    dev = torch.device("cuda:0")
    T = 100
    M = _model(hip, T)
    B = 4096
    assert hip.solve_workspace_bytes(M, B) > hip.STATIC_WORKSPACE_BYTES
    x0_h, us_h = _batch(B, T, 5)
    delta = torch.full((B,), 2.0, dtype=torch.float64, device=dev)
    k = torch.zeros((B, T, 1), dtype=torch.float64, device=dev)
    K = torch.zeros((B, T, 1, 5), dtype=torch.float64, device=dev)
    with pytest.raises(hip.IlqrHipError, match="workspace"):
        hip.solve_static(M, hip.Opts(), torch.tensor(x0_h, device=dev), torch.tensor(us_h, device=dev), delta, k, K,
                         _outputs(B, T, dev))


def test_static_workspace_solve_can_be_captured_in_a_graph_and_replayed(hip):
    """What the entry point is for: after one warm-up call it can be recorded in a HIP graph (recording fails on
    any allocation or synchronisation inside the call) and replayed on new inputs written into the same
    tensors; every replay equals ilqr_solve_batched on those inputs.  All calls on a GPU share the library's
    workspace, so replays (and any other caller of this entry point) must not overlap: they are serialised on
    one stream here."""
    dev = torch.device("cuda:0")
    T, B = 20, 400
    M = _model(hip, T)
    O = hip.Opts(max_iter=12)
    f64 = torch.float64
    x0 = torch.empty((B, 5), dtype=f64, device=dev)
    us0 = torch.empty((B, T, 1), dtype=f64, device=dev)
    delta = torch.empty(B, dtype=f64, device=dev)
    k = torch.empty((B, T, 1), dtype=f64, device=dev)
    K = torch.empty((B, T, 1, 5), dtype=f64, device=dev)
    out = _outputs(B, T, dev)

    def load(seed):
        x0_h, us_h = _batch(B, T, seed)
        x0.copy_(torch.tensor(x0_h))
        us0.copy_(torch.tensor(us_h))
        delta.fill_(2.0)
        k.zero_()
        K.zero_()

    load(20)
    hip.solve_static(M, O, x0, us0, delta, k, K, out)  # warm-up: the per-device set-up happens here
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        hip.solve_static(M, O, x0, us0, delta, k, K, out)
    for seed in (21, 22):
        load(seed)
        graph.replay()
        torch.cuda.synchronize()
        ref = hip.solve(M, O, x0.clone(), us0.clone())
        for key in ("xs", "us", "J", "mu", "iters", "status"):
            assert torch.equal(out[key], ref[key]), (seed, key)
        assert torch.equal(delta, ref["delta"]) and torch.equal(k, ref["k"]) and torch.equal(K, ref["K"])
