"""The Lightning route of INTEGRATION.md section 1 (`pl.Trainer(...).test(model, dataloaders, ckpt_path)`, reference
eval_MoCoDAD.py:24-38) exercised against a Lightning-SHAPED stand-in: pytorch_lightning is not installed in this image, so the
tests install a module of that name whose LightningModule has what the real base class has and the product's own fallback
base does not -- a READ-ONLY `device` property maintained by `.to()`, `log()` that reports to the attached trainer, the epoch
hooks, `save_hyperparameters` filling `hparams` -- and a 20-line Trainer.test that loads a Lightning checkpoint
({"state_dict": ...}) and drives on_test_epoch_start / test_step / on_test_epoch_end.  mocodad_amd.models.mocodad is then
re-imported so that MoCoDAD subclasses THAT base (models/mocodad.py:27-29).

CPU test: import, construction, checkpoint load, hook wiring, log -- with the scoring call replaced by a stand-in (there is
no CPU scoring path).  GPU test: the whole route with the real kernels, equal to the Trainer-free loop bit for bit."""
import importlib
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import golden_weights, make_args


def _fake_lightning():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.__dev = torch.device("cpu")
            self.trainer = None
            self.hparams = None

        @property
        def device(self):                       # read-only, like lightning's _DeviceDtypeModuleMixin
            return self.__dev

        def _apply(self, fn, *a, **k):
            out = super()._apply(fn, *a, **k)
            for p in self.parameters():
                self.__dev = p.device
                break
            return out

        def log(self, name, value, **kw):
            if self.trainer is None:
                raise RuntimeError("self.log() outside of a Trainer loop")
            self.trainer.callback_metrics[name] = float(value)

        def save_hyperparameters(self, *args, **kw):
            self.hparams = args[0] if args else None

        def on_test_epoch_start(self):
            self.hook_calls = getattr(self, "hook_calls", 0) + 1

        def on_validation_epoch_start(self):
            pass

    class Trainer:
        def __init__(self, accelerator="cpu", devices=1, **kw):
            self.device = torch.device("cuda:0" if accelerator in ("gpu", "cuda") else "cpu")
            self.callback_metrics = {}

        def test(self, model, dataloaders=None, ckpt_path=None):
            if ckpt_path is not None:
                model.load_state_dict(torch.load(ckpt_path, map_location="cpu", weights_only=False)["state_dict"])
            model.trainer = self
            model.to(self.device).eval()
            with torch.no_grad():
                model.on_test_epoch_start()
                for i, batch in enumerate(dataloaders):
                    model.test_step(batch, i)
                model.on_test_epoch_end()
            return [dict(self.callback_metrics)]

    pl.LightningModule, pl.Trainer = LightningModule, Trainer
    return pl


@pytest.fixture
def lightning_mocodad():
    """mocodad_amd.models.mocodad re-imported with a Lightning-shaped `pytorch_lightning` in sys.modules; restored afterwards."""
    import mocodad_amd.models.mocodad as mod
    assert "pytorch_lightning" not in sys.modules or getattr(sys.modules["pytorch_lightning"], "__file__", None) is None
    saved = sys.modules.get("pytorch_lightning")
    pl = _fake_lightning()
    sys.modules["pytorch_lightning"] = pl
    try:
        mod = importlib.reload(mod)
        assert issubclass(mod.MoCoDAD, pl.LightningModule)
        yield mod, pl
    finally:
        if saved is None:
            sys.modules.pop("pytorch_lightning", None)
        else:
            sys.modules["pytorch_lightning"] = saved
        importlib.reload(mod)


def _dataset(tmp_path):
    from mocodad_amd.data import synthetic
    data, trans, meta, frames, gts = synthetic.make_dataset(n_clips=2, frames_per_clip=40, persons_per_clip=2, num_transform=2)
    synthetic.write_gt(str(tmp_path / "gt"), gts)
    return (data, trans, meta, frames), gts, synthetic.batches((data, trans, meta, frames), 100)


def _ckpt(tmp_path):
    sd, cfg = golden_weights("inject")
    path = str(tmp_path / "epoch=0.ckpt")
    torch.save({"state_dict": sd, "epoch": 0, "pytorch-lightning_version": "2.0.0"}, path)
    return path, sd, cfg


def test_lightning_route_cpu_wiring(tmp_path, lightning_mocodad):
    mod, pl = lightning_mocodad
    from oracle import mocodad_oracle as O
    path, sd, cfg = _ckpt(tmp_path)
    tensors, gts, loader = _dataset(tmp_path)
    args = make_args(cfg, noise_steps=4, n_generated_samples=2, gt_path=str(tmp_path / "gt"), num_transform=2, dataset_choice="HR-STC",
                     pad_size=-1, filter_kernel_size=3, frames_shift=2, save_tensors=False)
    m = mod.MoCoDAD(args)
    assert m.hparams is args and m.device == torch.device("cpu")
    with pytest.raises(AttributeError):
        m.device = torch.device("cpu")              # the real base's property has no setter either: the module must not need one
    assert set(m.state_dict().keys()) == set(sd.keys())
    # no CPU scoring path exists (and must not): forward on a cpu module raises
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.forward(loader[0])
    # the hooks / collation / AUC / log wiring with the scoring call replaced by a deterministic stand-in
    rng = np.random.default_rng(5)
    fake_scores = []

    def fake_forward(batch):
        s = torch.from_numpy(rng.gamma(2.0, 0.05, size=batch[0].shape[0]).astype(np.float32))
        fake_scores.append(s.numpy())
        return [s, batch[0], batch[1], batch[2], batch[3]]
    m.forward = fake_forward
    tr = pl.Trainer(accelerator="cpu")
    res = tr.test(m, dataloaders=loader, ckpt_path=path)
    assert m.hook_calls == 1                       # super().on_test_epoch_start() reached the base class
    for k, v in sd.items():                        # the checkpoint's state_dict was loaded verbatim
        assert torch.equal(m.state_dict()[k], v), k
    auc_ref, _, _ = O.post_processing(np.concatenate(fake_scores), tensors[1].numpy(), tensors[2].numpy(), tensors[3].numpy(), gts,
                                      num_transform=2, pad_size=-1, filter_kernel_size=3, frames_shift=2)
    assert abs(res[0]["AUC"] - auc_ref) < 1e-12


@pytest.mark.gpu
def test_lightning_route_gpu_equals_trainer_free_loop(tmp_path, lightning_mocodad):
    mod, pl = lightning_mocodad
    path, sd, cfg = _ckpt(tmp_path)
    tensors, gts, loader = _dataset(tmp_path)
    kw = dict(noise_steps=4, n_generated_samples=2, gt_path=str(tmp_path / "gt"), num_transform=2, dataset_choice="HR-STC",
              pad_size=-1, filter_kernel_size=3, frames_shift=2, save_tensors=False)
    torch.manual_seed(1)
    m = mod.MoCoDAD(make_args(cfg, **kw))           # random init; the checkpoint comes through Trainer.test
    tr = pl.Trainer(accelerator="gpu", devices=1)
    res = tr.test(m, dataloaders=loader, ckpt_path=path)
    assert m.device.type == "cuda"
    # the Trainer-free loop of eval_MoCoDAD.py on the same weights
    m2 = mod.MoCoDAD(make_args(cfg, **kw))
    m2.trainer = pl.Trainer()
    m2.load_state_dict(sd)
    m2.to("cuda:0")
    m2.on_test_epoch_start()
    for i, b in enumerate(loader):
        m2.test_step(b, i)
    auc2 = m2.on_test_epoch_end()
    assert res[0]["AUC"] == auc2 and np.array_equal(m.last_scores, m2.last_scores)
    assert 0.0 <= auc2 <= 1.0
