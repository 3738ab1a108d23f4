import sys, time, os, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, sklearn.metrics
from mocodad_amd.data import synthetic
from mocodad_amd.models.mocodad import MoCoDAD
from mocodad_amd.utils.argparser import load_config
from mocodad_amd.utils.eval_utils import post_process_scores
args = load_config("configs/hr_avenue_test.yaml")
data, trans, meta, frames, gts = synthetic.make_dataset(n_clips=64, frames_per_clip=200, seg_len=6, num_transform=5, seed=999)
d = tempfile.mkdtemp(); synthetic.write_gt(d, gts); args.gt_path = d
m = MoCoDAD(args).to("cuda:0"); m.dataset_name = "synthetic"
out = np.random.default_rng(0).gamma(2.0, 0.05, size=data.shape[0]).astype(np.float32)
tr, me, fr = trans.numpy(), meta.numpy(), frames.numpy()
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    auc = m.post_processing(out, None, tr, me, fr)
    torch.cuda.synchronize(); print(f"device path call {i}: {time.perf_counter()-t0:.4f}s auc {auc:.6f}", flush=True)
asm = m._frame_assembler()
dtr, dme, dfr, dout = torch.from_numpy(tr).cuda(), torch.from_numpy(me).cuda(), torch.from_numpy(fr).cuda(), torch.from_numpy(out).cuda()
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pds = asm(dout, dtr, dme, dfr)
    t1 = time.perf_counter()
    a = sklearn.metrics.roc_auc_score(asm.gt, pds)
    print(f"device-resident inputs: assembly {t1-t0:.4f}s + roc_auc {time.perf_counter()-t1:.4f}s  ({len(pds)} frames)", flush=True)
g2, masks = m._gt_and_masks()
t0 = time.perf_counter()
p2, gt = post_process_scores(out, tr, me, fr, g2, num_transform=5, pad_size=args.pad_size, filter_kernel_size=args.filter_kernel_size, frames_shift=args.frames_shift, masks=masks)
print(f"host NumPy path: {time.perf_counter()-t0:.4f}s  max |pds diff| {np.abs(p2-pds).max():.2e}")
