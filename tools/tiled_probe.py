import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from helpers import golden_weights, make_args
from oracle import mocodad_oracle as O
from mocodad_amd.models.mocodad import MoCoDAD
def run(strategy, seg_len, ci):
    _, cfg = golden_weights("inject")
    torch.manual_seed(5)
    m = MoCoDAD(make_args(cfg, conditioning_strategy=strategy, seg_len=seg_len, conditioning_indices=ci, noise_steps=4, n_generated_samples=3))
    gen = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=gen) * 0.1)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=gen) + 0.5)
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=gen) + 0.5)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=gen) * 0.1)
        last = m.model.st_gcnnsu3[-1]
        last.tcn[0].weight.mul_(0.25); last.residual[0].weight.mul_(0.25)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to("cuda:0")
    B, S, ns = 5, 3, 4
    data = torch.randn(B, 2, seg_len, 17, generator=gen).clamp_(-3, 3)
    Tx = m.n_frames_corrupt
    noise = torch.randn(S, ns - 1, B, 2, Tx, 17, generator=gen)
    batch = [data, torch.zeros(B), torch.zeros(B, 4), torch.zeros(B, seg_len)]
    out = m.forward(batch, aggr_strategy="all", return_="all", noise=noise)
    with torch.no_grad():
        p_ref, corrupt = O.reverse_diffusion(sd, data, noise, noise_steps=ns, strategy=strategy, conditioning_indices=ci)
        l_ref = O.window_losses(p_ref, corrupt)
    ep = np.abs(out[1].cpu().numpy() - p_ref.transpose(0, 1).numpy()).max()
    el = np.abs(out[0].cpu().numpy() - l_ref.t().numpy()).max()
    sc = m.scorer(); sc.set_option("generic_unet", 1)
    out2 = m.forward(batch, aggr_strategy="all", return_="all", noise=noise)
    eg = np.abs(out2[0].cpu().numpy() - l_ref.t().numpy()).max()
    print(strategy, seg_len, ci, "T_u", m.input_n_frames, "max err pose %.3e loss %.3e (generic kernel loss err %.3e)" % (ep, el, eg), flush=True)
for a in [("inject", 32, 2), ("concat", 24, [0, 1, 2, 3]), ("concat", 13, [0, 1, 2]), ("inject", 26, 2), ("concat", 20, [0,1]), ("inbetween_imp", 30, 3), ("no_condition", 17, None), ("concat", 32, [28, 29, 30, 31])]:
    run(*a)
