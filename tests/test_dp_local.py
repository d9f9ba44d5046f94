"""BucketedGradReducer without a process group (world size 1): the bucket bookkeeping on its own -- gradients gathered
into the flat buckets with one multi-tensor launch per bucket, accumulation micro-steps, set_to_none callers, parameters
without gradient."""
import torch

from caduceus_amd.dp import BucketedGradReducer


def _model():
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                            torch.nn.Linear(16, 4))
    m.unused = torch.nn.Parameter(torch.ones(5))  # never reaches the loss
    return m


def _ref_grads(m, xs):
    ref = {k: torch.zeros_like(p) for k, p in m.named_parameters()}
    for x in xs:
        gs = torch.autograd.grad(m(x).square().mean(), [p for k, p in m.named_parameters() if k != "unused"])
        for (k, _), g in zip([(k, p) for k, p in m.named_parameters() if k != "unused"], gs):
            ref[k] += g
    return ref


def test_gather_accumulate_and_set_to_none():
    m = _model()
    red = BucketedGradReducer(m.parameters(), bucket_bytes=300)
    assert len(red.buckets) >= 2
    xs = [torch.randn(3, 8, generator=torch.Generator().manual_seed(s)) for s in range(3)]
    # 1. plain step
    red.zero_grad()
    assert all(p.grad is None for p in m.parameters())
    m(xs[0]).square().mean().backward()
    red.finish()
    ref = _ref_grads(m, xs[:1])
    for k, p in m.named_parameters():
        assert p.grad.data_ptr() == red._views[id(p)].data_ptr()  # the gradient lives in its bucket
        torch.testing.assert_close(p.grad, ref[k], rtol=0, atol=0)
    # 2. a new step OVERWRITES (stale bucket content does not leak), incl. the parameter that gets no gradient
    red.buckets[0].fill_(7.0)
    red.zero_grad()
    m(xs[1]).square().mean().backward()
    red.finish()
    ref = _ref_grads(m, xs[1:2])
    for k, p in m.named_parameters():
        torch.testing.assert_close(p.grad, ref[k], rtol=0, atol=0)
    # 3. accumulation micro-steps under no_sync add up
    red.zero_grad()
    with red.no_sync():
        m(xs[0]).square().mean().backward()
        m(xs[1]).square().mean().backward()
    m(xs[2]).square().mean().backward()
    red.finish()
    ref = _ref_grads(m, xs)
    for k, p in m.named_parameters():
        torch.testing.assert_close(p.grad, ref[k], rtol=1e-6, atol=1e-7)
    # 4. a caller that clears gradients itself
    m.zero_grad(set_to_none=True)
    m(xs[2]).square().mean().backward()
    red.finish()
    ref = _ref_grads(m, xs[2:])
    for k, p in m.named_parameters():
        assert p.grad.data_ptr() == red._views[id(p)].data_ptr()
        torch.testing.assert_close(p.grad, ref[k], rtol=0, atol=0)
