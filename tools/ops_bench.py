import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import articulated_pose_amd
from articulated_pose_amd import tf_ops, _lib
from articulated_pose_amd.tf_ops.tf_sampling import farthest_point_sample_gather
from articulated_pose_amd.synthetic import make_batch
dev='cuda:0'
B=32
P=torch.from_numpy(make_batch(0,B,N=1024,K=3)['P']).to(dev)
_,l1=farthest_point_sample_gather(512,P)
_,l2=farthest_point_sample_gather(128,l1)
f1=torch.randn(B,512,128,device=dev)
def run():
    if os.environ.get('FUSED')=='1':
        _,_,g1=tf_ops.query_ball_group_xyz(0.2,64,P,l1); idx2,_,g2=tf_ops.query_ball_group_xyz(0.4,64,l1,l2)
        return g1,g2,tf_ops.group_point(f1,idx2)
    if os.environ.get('MULTI')=='1':
        (idx1,_),(idx2,_)=tf_ops.query_ball_point_multi([(0.2,64,P,l1),(0.4,64,l1,l2)])
        if os.environ.get('GMULTI')=='1':
            g1,g2=tf_ops.group_point_multi([(P,idx1),(l1,idx2)]); return g1,g2,tf_ops.group_point(f1,idx2)
        return tf_ops.group_point(P,idx1),tf_ops.group_point(l1,idx2),tf_ops.group_point(f1,idx2)
    idx1,_=tf_ops.query_ball_point(0.2,64,P,l1); g1=tf_ops.group_point(P,idx1)
    idx2,_=tf_ops.query_ball_point(0.4,64,l1,l2); g2=tf_ops.group_point(l1,idx2); g3=tf_ops.group_point(f1,idx2)
    return g1,g2,g3
s=torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): run()
s.synchronize()
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g,stream=s): out=run()
with torch.cuda.stream(s):
    for _ in range(400): g.replay()          # loaded clock
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): g.replay()
    e1.record()
s.synchronize()
us=e0.elapsed_time(e1)/200*1e3
tb=(151552+536576+40960+137216+4489216)*B
print('graph of 5 op launches: %.1f us  -> %.1f GB/s = %.3f of 8 TB/s'%(us,tb/us/1e3,tb/us/1e3/8000))
_lib.profile_start(lead=50); run(); _lib.profile_stop()
_lib.profile_start(lead=50); run(); rec=_lib.profile_stop()
for n,a,ms in rec: print(n, round(ms*1e3,1),'us')
