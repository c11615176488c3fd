"""Debug helper: what observation does the in-kernel network of the rigid-body policy kernel (quad mapping) SEE?  A network that
copies five of its inputs to its outputs (ReLU kept linear by a bias of 10) is run four times to recover all 18 entries."""
import sys, torch
sys.path.insert(0, '.')
from rl_on_manifold_amd import BatchedAtacomEnv, MlpPolicy
DEV = 'cuda:0'
import os
B, T = int(os.environ.get('DBG_B', '32')), int(os.environ.get('DBG_T', '3'))
MODE, LANES = os.environ.get('DBG_MODE', 'rigid_body'), int(os.environ.get('DBG_LANES', '4'))
torch.set_printoptions(precision=5, linewidth=200, sci_mode=False)
seen = torch.zeros((T, B, 18), device=DEV)
ref = None
for j0 in (0, 5, 10, 13):
    W1 = torch.zeros(64, 18); W1[:18, :18] = torch.eye(18)
    W2 = torch.eye(64); W3 = torch.zeros(5, 64)
    for k in range(5): W3[k, j0 + k] = 1.0
    pol = MlpPolicy(W1, torch.full((64,), 10.0), W2, torch.zeros(64), W3, torch.full((5,), -10.0), std=torch.zeros(5))
    env = BatchedAtacomEnv('iiwa', B, device=DEV, dynamics_mode=MODE, lanes_per_env=LANES)
    # a fixed action sequence would decouple the runs; here the actions ARE the observations, so runs differ after step 0:
    # only compare step 0 and step 1 inputs against the run's own observations
    out = env.rollout_policy(pol, T, noise=None)
    for t in range(T):
        d = out['action'][t] - out['obs'][t][:, j0:j0 + 5]
        bad = torch.nonzero(d.abs().amax(1) > 1e-4).flatten().tolist()
        print('inputs %2d..%2d, t %d: envs whose network input differs from the observation written out: %s' % (j0, j0 + 4, t, bad[:16]))
        if t == 1 and bad:
            e = bad[0]
            print('   env %d: written obs %s' % (e, out['obs'][t][e, j0:j0 + 5].cpu().numpy()))
            print('   env %d: network saw %s' % (e, out['action'][t][e].cpu().numpy()))
            for other in (e & ~3, e - 1, e + 1):
                if 0 <= other < B:
                    print('   env %d wrote    %s   (t-1: %s)' % (other, out['obs'][t][other, j0:j0 + 5].cpu().numpy(), out['obs'][t - 1][e, j0:j0 + 5].cpu().numpy()))
