"""The reference's single-problem class on a few of its registered env ids (examples/ilqr/*.py style):
    PYTHONPATH=. python scripts/example_env_ids.py        (needs an MI355X)"""
import numpy as np
from trajopt_amd import envs
from trajopt_amd.ilqr import iLQR
for eid, T in (("Cartpole-TO-v1", 100), ("QuadPendulum-TO-v0", 60), ("Robot-TO-v0", 80), ("LQR-TO-v1", 60)):
    env = envs.make(eid)
    env.seed(1337)
    x0 = env.init()[0]
    alg = iLQR(env, nb_steps=T, init_state=x0)
    trace = alg.run(nb_iter=5, verbose=False)
    print(eid, "trace:", ["%.4g" % v for v in trace], "K", alg.ctl.K.shape, flush=True)
    assert np.all(np.isfinite(trace)) and trace[-1] <= trace[0]
print("examples ok")
