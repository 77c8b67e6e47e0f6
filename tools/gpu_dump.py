import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from marinevehiclereinforcementlearning_amd import _lib, params as P
name, dof, n_sub = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
g = np.load(os.path.join(REPO, "tests", "golden", name))
n_env, n_steps = g["actions"].shape[:2]
cfg = P.make_config(P.MODEL_ROV6 if dof == 6 else P.MODEL_ROV3, n_env, n_substeps=n_sub, fixed_setpoint=bool(g["fixedSp"]),
                    auto_reset=False, max_steps=10 ** 9, use_flow=False)
h = _lib.Handle(cfg)
h.enable_aux(True)
npos = 3 if dof == 6 else 2
init = np.concatenate([g["path"].reshape(n_env, 2 * npos), g["sp0"][:, npos:]], axis=1)
h.reset(init=init)
S, A = [], []
for s in range(n_steps):
    h.step(g["actions"][:, s])
    S.append(h.get_state().copy()); A.append(h.get_aux().copy())
np.savez(os.path.join(REPO, "gpurun_out", "dump_" + name), states=np.array(S), aux=np.array(A))
