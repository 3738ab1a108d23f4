"""YAML -> argparse.Namespace, with the derived paths the reference's utils/argparser.py:4-43 adds."""
import argparse
import os

import yaml


def init_args(args: argparse.Namespace) -> argparse.Namespace:
    args.gt_path = args.test_path
    root = os.path.join(args.data_dir, "pose")
    args.pose_path = {"train": os.path.join(root, "training/tracked_person/"),
                      "test": os.path.join(root, "testing/tracked_person/"),
                      "validation": os.path.join(root, "validating/tracked_person/")}
    if getattr(args, "validation", False):
        args.gt_path = os.path.join(args.data_dir, "validating", "test_frame_mask")
    args.ckpt_dir = os.path.join(args.exp_dir, args.dataset_choice, args.dir_name)
    if getattr(args, "create_experiment_dir", False):
        os.makedirs(args.ckpt_dir, exist_ok=True)
    return args


def load_config(path: str) -> argparse.Namespace:
    with open(path) as f:
        return init_args(argparse.Namespace(**yaml.load(f, Loader=yaml.FullLoader)))
