import os, sys
sys.path.insert(0, ".")
import numpy as np
from tests import util
from tests.test_gpu_parity import _states_with_slots
task = sys.argv[1] if len(sys.argv) > 1 else "reach"
qpos, qvel = _states_with_slots(task, [16], 256, seed=66, cube_near_gripper=False, action_mode=0)
n = len(qpos)
res = {}
for m in ("0", "4"):
    os.environ["LCR_COOP_MAX"] = m
    sim, o = util.make_pair(task, n, auto_reset=False, max_episode_steps=0, action_mode="joint")
    o.reset(seeds=np.arange(n)); sim.reset(seeds=np.arange(n))
    o.qpos[:] = qpos; o.qvel[:] = qvel
    util.sync_oracle_to_f32(o, carry=False)
    util.push_state(sim, o, carry=False)
    rng = np.random.default_rng(3)
    a = (0.3 * rng.uniform(-1, 1, (n, sim.action_dim))).astype(np.float32)
    sim.step(a)
    st = sim.get_state()
    res[m] = (st["qpos"].T.copy(), sim.active_mask.numpy().copy(), sim.max_sweeps.numpy().copy())
    sim.close()
d = np.abs(res["0"][0] - res["4"][0]).max(axis=1)
print("envs", n, "differing >1e-5:", int((d > 1e-5).sum()), "max", d.max())
for e in np.argsort(-d)[:12]:
    m0, m4 = res["0"][1][e], res["4"][1][e]
    print(f"env {e:4d} (wave {e//64}, lane {e%64}) dq {d[e]:.3e} mask0 {m0:#08x} mask4 {m4:#08x} sweeps {res['0'][2][e]} {res['4'][2][e]}  arm dq {np.abs(res['0'][0][e,:6]-res['4'][0][e,:6]).max():.2e} cube dq {np.abs(res['0'][0][e,6:]-res['4'][0][e,6:]).max():.2e}")
# coupled lanes: bits 12, 13 (finger on cube), 16 with on cube unknown
cp = ((res["4"][1] >> 12) & 3) != 0
print("lanes with finger on cube:", np.nonzero(cp)[0][:40].tolist())
for e in (93, 175, 105, 22, 30):
    print("patient", e, "sweeps", res["0"][2][e], hex(int(res["4"][2][e])), "dq", d[e])
big = np.nonzero(d > 1e-5)[0]
print("all envs differing > 1e-5:", [(int(e), int(e) // 64, int(e) % 64, float(f"{d[e]:.2e}"), hex(int(res["4"][2][e]))) for e in big])
c4 = np.nonzero(res["4"][2] >= 30)[0]
print("envs with sweeps >= 30 under coop:", c4.tolist())
