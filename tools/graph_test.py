import sys, time
sys.path.insert(0,'greenlight-gym2_amd')
import torch, numpy as np
from gl_gym_amd.tomato_env import TomatoVecEnv
from gl_gym_amd.utils import synthetic_weather
B=65536
w=synthetic_weather(35040); starts=np.arange(0,35040-5760-60,96)
env=TomatoVecEnv(B, weather=w, dtype='float32', season_length=60, start_rows=starts.tolist(), start_days=(starts/96.0).tolist(), seed=666)
env.reset_tensor()
dev=env.device
gen=torch.Generator(device=dev).manual_seed(1)
acts=[torch.rand(B,6,generator=gen,device=dev)*2-1 for _ in range(8)]
static_a=torch.zeros(B,6,device=dev)
def seq():
    env.action_t.copy_(static_a)
    env._launch_step(raw_control=False)
    env._launch_obs(env.obs_t)
    env._launch_reset(env.done_t)
    env._launch_obs(env.obs_t, env.done_t, env.term_obs_t)
# eager timing
for i in range(5): static_a.copy_(acts[i%8]); seq()
torch.cuda.synchronize(); t0=time.perf_counter()
for i in range(50): static_a.copy_(acts[i%8]); seq()
torch.cuda.synchronize(); print('eager  %.4f ms/step'%((time.perf_counter()-t0)/50*1e3))
# graph
s=torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    seq()
torch.cuda.current_stream().wait_stream(s)
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    seq()
for i in range(5): static_a.copy_(acts[i%8]); g.replay()
torch.cuda.synchronize(); t0=time.perf_counter()
for i in range(50): static_a.copy_(acts[i%8]); g.replay()
torch.cuda.synchronize(); print('graph  %.4f ms/step'%((time.perf_counter()-t0)/50*1e3))
print('timestep', int(env.timestep_t[0]), 'fail', env.metrics()['n_ode_fail'])
