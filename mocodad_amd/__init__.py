"""mocodad_amd — MI355X-native implementation of MoCoDAD's anomaly-scoring hot path.

Public surface (mirrors the reference's modules so an eval_MoCoDAD.py-shaped driver is a drop-in):
  mocodad_amd.models.mocodad.MoCoDAD          <- models/mocodad.py
  mocodad_amd.utils.diffusion_utils.Diffusion <- utils/diffusion_utils.py
  mocodad_amd.engine.HipScorer                   thin driver over the C ABI (include/mocodad_hip.h)
"""
__version__ = "0.1.0"
