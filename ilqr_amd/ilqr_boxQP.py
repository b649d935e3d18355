"""Box-constrained QP (mirror of ilqr/ilqr_boxQP.py) on the GPU.

``boxQP(H, g, lower, upper, x0=None, **opts)`` keeps the reference's signature, option
names, defaults, result codes and return tuple.  H may also be a batch [B, n, n]
(with g, lower, upper, x0 of shape [B, n]); all QPs of a batch run in one launch,
one QP per lane.
"""
import numpy as np

from . import _hip

RESULT_TEXT = {
    -1: 'Hessian is not positive definite',
    0: 'No descent direction found',
    1: 'Maximum main iterations exceeded',
    2: 'Maximum line-search iterations exceeded',
    3: 'No bounds, returning Newton point',
    4: 'Improvement smaller than tolerance',
    5: 'Gradient norm smaller than tolerance',
    6: 'All dimensions are clamped',
}


def boxQP(H, g, lower, upper, x0=None, *args, **kwargs):
    """Minimize 0.5*x'*H*x + x'*g  s.t. lower<=x<=upper with projected Newton steps.

    Returns (x, result, Hfree, free): solution [n], result code (RESULT_TEXT), Cholesky
    factor of the free sub-space [n_free, n_free] (all ones [n, n] if the solve ended
    before any factorisation, as in the reference) and the free mask [n].
    For a batch: x [B, n], result [B], Hfree [B, n, n] with the factor in the leading
    n_free block, free [B, n].
    """
    import torch
    opts = {'maxIter': 100, 'minGrad': 1e-8, 'minRelImprove': 1e-8, 'stepDec': 0.6, 'minStep': 1e-22,
            'Armijo': 0.1, 'verbose': 0}  # ilqr_boxQP.py:55
    for key, value in kwargs.items():
        opts[str(key)] = value
    H = np.asarray(H, dtype=np.float64)
    single = H.ndim == 2
    n = H.shape[-1]
    Hb = H.reshape(-1, n, n)
    B = Hb.shape[0]

    def vec(a):
        return np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64).reshape(-1, n), (B, n)))
    gb, lb, ub = vec(g), vec(lower), vec(upper)
    if x0 is not None:
        assert np.asarray(x0).shape[-1] == n, \
            "boxQP: dimension mismatch between H{} and x{}!".format(str(n) + "x" + str(n), str(n))
        x0b = vec(x0)
    else:
        assert np.isfinite(np.hstack([lb, ub])).all(), "boxQP: invalid inputs, inputs need to be finite!"
        x0b = None
    if not torch.cuda.is_available():
        raise _hip.IlqrHipError("no GPU visible: boxQP has no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    O = _hip.Opts(qp_max_iter=int(opts['maxIter']), qp_min_grad=opts['minGrad'],
                  qp_min_rel_improve=opts['minRelImprove'], qp_step_dec=opts['stepDec'],
                  qp_min_step=opts['minStep'], qp_armijo=opts['Armijo'])

    def t(a):
        return None if a is None else torch.tensor(a, device=dev)
    out = _hip.boxqp_batched(t(np.ascontiguousarray(Hb)), t(gb), t(lb), t(ub), t(x0b), O)
    x = out["x"].cpu().numpy()
    result = out["result"].cpu().numpy().astype(np.int64)
    free = out["free"].cpu().numpy().astype(bool)
    Hff = out["Hff"].cpu().numpy()
    dims = out["hff_dim"].cpu().numpy()
    if (result == -99).any():
        # np.linalg.eigvals / scipy cholesky raise on these inputs in the reference
        raise np.linalg.LinAlgError("boxQP: H is not finite or a free sub-block is not positive definite")
    if opts['verbose'] > 0:
        for b in range(B):
            print('RESULT({}): {}.'.format(result[b], RESULT_TEXT[int(result[b])]))
    if single:
        d = int(dims[0])
        return x[0], int(result[0]), np.tril(Hff[0][:d, :d]) if d < n or not np.all(Hff[0] == 1.0) else Hff[0], free[0]
    return x, result, Hff, free
