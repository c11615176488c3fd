"""Step server (atacom_server_start / _submit / _stop; VERDICT r3 item 5 -- an experiment whose measured outcome is in
profiles/r04_step_server.md): a persistent launch with the state in registers serving one env step per submission must
produce exactly what atacom_step produces, refuse what it cannot hold resident, and can never hang the device."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]
DEV = 'cuda:0'


def _pair(name, B, **kw):
    from rl_on_manifold_amd import BatchedAtacomEnv
    mk = lambda: BatchedAtacomEnv(name, B, device=DEV, auto_reset=True, random_init=(name != 'iiwa'), seed=3, horizon=9, **kw)   # noqa: E731
    a, b = mk(), mk()
    if name != 'circle':
        st = a.get_state()
        nq = a.dims['q']
        g = torch.Generator(device=DEV).manual_seed(5)
        init = torch.zeros((B, a.init_state_dim), device=DEV)
        init[:, :nq] = st[:, :nq] + 0.04 * torch.randn((B, nq), device=DEV, generator=g)
        init[:, 2 * nq:] = st[:, 2 * nq + a.dims['g']:2 * nq + a.dims['g'] + 6]
        a.reset(state=init); b.reset(state=init)
    else:
        a.reset(); b.reset()
    a.get_constraints_logs(); b.get_constraints_logs()
    return a, b


@pytest.mark.parametrize('transport', ['kernel', 'stream_ops'])
@pytest.mark.parametrize('name,lanes,chart', [('iiwa', 4, 'reference'), ('iiwa', 1, 'reference'), ('iiwa', 4, 'canonical'),
                                              ('planar', 4, 'reference'), ('planar', 1, 'canonical'), ('circle', 1, 'reference')])
def test_step_server_equals_atacom_step(name, lanes, chart, transport):
    """T submissions through the server == T atacom_step calls on a twin handle, bit for bit: observations, rewards, flags
    (auto-resets and device-side random starts included), the state written back at stop, the constraint statistics."""
    B, T = 1000, 25
    a, b = _pair(name, B, lanes_per_env=lanes, chart_mode=chart)
    k = a.dims['null']
    g = torch.Generator(device=DEV).manual_seed(1)
    acts = torch.rand((T, B, k), device=DEV, generator=g) * 2.4 - 1.2
    ref = []
    for t in range(T):
        a.step_into(acts[t], a._obs, a._reward, a._absorbing, a._last)
        ref.append(tuple(x.clone() for x in (a._obs, a._reward, a._absorbing, a._last)))
    buf = torch.empty((B, k), device=DEV)
    got = []
    with b.serve(buf, max_steps=T, timeout_s=3.0, transport=transport) as srv:
        with pytest.raises(Exception, match='serving'):
            b.get_state()                                       # the state is in the launch
        for t in range(T):
            buf.copy_(acts[t])
            srv.submit()
            got.append((srv.obs.clone(), srv.reward.clone(), srv.absorbing.clone(), srv.last.clone()))
        with pytest.raises(Exception, match='used up'):
            srv.submit()
    for t in range(T):
        for i, (x, y) in enumerate(zip(ref[t], got[t])):
            assert torch.equal(x, y), (t, i, (x.double() - y.double()).abs().max().item(), int((x != y).sum()))
    assert any(r[3].any() for r in ref)                        # episodes ended on the way (horizon 9)
    assert torch.equal(a.get_state(), b.get_state())
    assert a.get_constraints_logs() == b.get_constraints_logs()
    # and the handle steps on normally afterwards
    o1 = a.step(acts[0])[0]
    o2 = b.step(acts[0])[0]
    assert torch.equal(o1, o2)


def test_step_server_with_a_torch_policy_between_submissions():
    """The loop it exists for: observation -> torch kernels (a small policy) -> action buffer -> submit, no host
    synchronisation per step; equals the same loop over atacom_step."""
    B, T = 2048, 40
    a, b = _pair('iiwa', B, lanes_per_env=4)                     # (the server runs the quad mapping: same summation order)
    W = torch.randn((18, 5), device=DEV, generator=torch.Generator(device=DEV).manual_seed(2)) * 0.5
    policy = lambda o: torch.tanh(o @ W) * 1.1                  # noqa: E731
    obs = a.reset(mask=torch.zeros(B, dtype=torch.uint8, device=DEV))
    ref = []
    for t in range(T):
        a.step_into(policy(obs), a._obs, a._reward, a._absorbing, a._last)
        obs = a._obs
        ref.append(obs.clone())
    buf = torch.empty((B, 5), device=DEV)
    got = []
    with b.serve(buf, max_steps=T, timeout_s=3.0) as srv:
        for t in range(T):
            buf.copy_(policy(srv.obs))
            srv.submit()
            got.append(srv.obs.clone())
    for t in range(T):
        assert torch.equal(ref[t], got[t]), t


def test_step_server_refusals_and_time_out():
    from rl_on_manifold_amd import BatchedAtacomEnv, _lib
    big = BatchedAtacomEnv('iiwa', 65536, device=DEV)           # 4096 wavefronts in the quad mapping: cannot all be resident
    buf = torch.empty((65536, 5), device=DEV)
    with pytest.raises(_lib.AtacomError, match='half of the device'):
        big.serve(buf, 10)
    big.close()
    f64 = BatchedAtacomEnv('planar', 256, device=DEV, dtype=torch.float64)
    with pytest.raises(_lib.AtacomError, match='float32'):
        f64.serve(torch.empty((256, 3), device=DEV, dtype=torch.float64), 10)
    dyn = BatchedAtacomEnv('iiwa', 256, device=DEV, dynamics_mode='rigid_body')
    with pytest.raises(_lib.AtacomError, match='kinematic'):
        dyn.serve(torch.empty((256, 5), device=DEV), 10)
    # a submission that never comes: the launch gives up after timeout_s, writes the state back, stop() reports it --
    # nothing hangs, and the handle is usable again
    env = BatchedAtacomEnv('planar', 512, device=DEV)
    before = env.get_state()
    import time
    srv = env.serve(torch.zeros((512, 3), device=DEV), max_steps=5, timeout_s=0.2)
    time.sleep(2.5)                                             # (the limit is counted in polls: 0.2 s nominal)
    with pytest.raises(_lib.AtacomError, match='timed out'):
        srv.stop()
    assert torch.equal(env.get_state(), before)
    env.step(torch.zeros((512, 3), device=DEV))
