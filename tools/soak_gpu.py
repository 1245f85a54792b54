import sys; sys.path.insert(0,'.')
import numpy as np, time
from gym_lowcostrobot_amd import VecSim
for task,mode in [('reach','joint'),('push','joint'),('lift','joint'),('pick_place','ee'),('stack','joint'),('push_loop','joint'),('lift','ee')]:
    n=65536
    sim=VecSim(task,n,observation_mode='state',action_mode=mode)
    act=sim.alloc_actions()
    early=0; succ=0
    t0=time.time()
    for t in range(500):
        sim.fill_random_actions(act,7,t); sim.step_device(act.ptr)
        if t%25==24:
            o=sim.outputs(); st=sim.get_state()
            # divergence guard fires => truncated while elapsed was < 50: detect via did_reset & ~terminated & elapsed pattern is hard; use state sanity instead
            assert np.isfinite(st['qpos']).all() and np.isfinite(st['qvel']).all()
            succ+=int(o['is_success'].sum())
    st=sim.get_state()
    print(task,mode,'ok; max|qvel arm| %.1f max|cube v| %.2f cube z range [%.3f, %.3f] max|q| %.2f successes(sampled) %d  %.1fs'%(np.abs(st['qvel'][:6]).max(), np.abs(st['qvel'][6:9]).max(), st['qpos'][8].min(), st['qpos'][8].max(), np.abs(st['qpos'][:6]).max(), succ, time.time()-t0))
    sim.close()
