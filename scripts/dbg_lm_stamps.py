import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rgbd_visualodometry_amd import capi
syn = capi.Synth(); sp = syn.params(seed=0)
bgr, depth, Twc, ts = syn.render(sp, 0, 10)
L = capi.load("rgbd_visualodometry_amd/csrc/build/libvo_hip_dbg.so")
def inv12(T):
    R=T[:9].reshape(3,3); t=T[9:]; return np.concatenate([R.T.ravel(), -R.T@t])
p = L.default_params(n_features=2000, max_frames=2); ctx = L.context(p)
ctx.upload(0, bgr[0], depth[0]); ctx.orb(0,1); k0,d0 = ctx.orb_fetch(0)
ok=k0['depth_raw']>0; z=k0['depth_raw'][ok]/5000.
pc=np.stack([(k0['x'][ok]-p.cx)*z/p.fx,(k0['y'][ok]-p.cy)*z/p.fy,z],1)
R0=Twc[0][:9].reshape(3,3); t0=Twc[0][9:]; pw=pc@R0.T+t0; nrm=pw-t0; nrm/=np.linalg.norm(nrm,axis=1,keepdims=True)
idx=np.arange(len(pw),dtype=np.int32); ctx.map_upsert(idx,pw,nrm,d0[ok],np.zeros(len(pw),np.uint8)); ctx.map_set_active(idx)
ctx.upload(1, bgr[8], depth[8]); ctx.orb(1,1)
tp = L.default_track_params(passes=1)
for rep in range(3):
    res, m = ctx.track(1, inv12(Twc[0]), tp)
    r = list(res.reserved)
    print("n_inl", res.n_ransac_inliers, "lm_iters", res.lm_iters, "passes", r[2], "| pass", r[0]*16, "serial", r[1]*16, "| loop", r[4]*16, "wave-reduce", r[5]*16, "lds+barriers", res.n_lm_inliers*16, "| kernel", r[6]*16)
