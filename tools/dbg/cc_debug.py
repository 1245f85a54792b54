import sys; sys.path.insert(0,'.')
import numpy as np
from tests import util
np.set_printoptions(precision=6, suppress=True, linewidth=200)
for iters in (0,1,2,10):
  for nsub in (1,):
    sim,o=util.make_pair("stack",4,auto_reset=False,max_episode_steps=0,pgs_iters=iters,n_substeps=nsub)
    o.reset(seeds=np.arange(4)); sim.reset(seeds=np.arange(4))
    o.qpos[:,6:9]=[0.25,0.25,0.015]; o.qpos[:,9:13]=[1,0,0,0]
    o.qpos[:,13:16]=[0.255,0.247,0.0445]
    yaw=np.array([0.0,0.3,0.0,0.5]); o.qpos[:,16]=np.cos(yaw/2); o.qpos[:,17:19]=0; o.qpos[:,19]=np.sin(yaw/2)
    o.qpos[2,13:15]=[0.25,0.25]
    o.qvel[:]=0
    util.sync_oracle_to_f32(o); util.push_state(sim,o)
    a=np.zeros((4,6),np.float32)
    o.step(a); sim.step(a)
    st=util.pull_state(sim)
    print('iters',iters,'nsub',nsub, 'diag',o.diag())
    print(' oracle qvel cubes', o.qvel[:,6:18])
    print(' hip    qvel cubes', st['qvel'][:,6:18])
    sim.close()
