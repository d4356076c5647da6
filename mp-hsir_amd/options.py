"""Command-line flags of the training script -- same names, types and defaults as the reference's
options.py (options.py:6-37), so its command lines (README.md:36,38) work unchanged:

    python train.py --epochs 100 --lr 2e-4 --batch_size 32 --data_type natural_scene ...

Like the reference, the namespace is built at import time (`from options import options as opt`);
unknown flags are tolerated so that importing this module under pytest/torchrun does not abort.
Additions (not present in the reference, all optional): --model, --precision, --steps_per_epoch, --graph,
--synthetic, --log_every, --allow_surrogate_clip, --all_sources, --resume.  Reference hazards kept on purpose: `--num_gpus type=list` turns "01"
into ['0','1'] (options.py:36) and `--classifier type=bool` treats any non-empty string as True.
"""
import argparse

_FLAGS = [
    # name, kwargs                                                                       (reference line)
    ("--cuda", dict(type=int, default=0)),                                               # :6
    ("--seed", dict(type=int, default=2024)),                                            # :7
    ("--epochs", dict(type=int, default=100, help="maximum number of epochs to train the total model.")),
    ("--batch_size", dict(type=int, default=32, help="Batch size to use per GPU")),
    ("--lr", dict(type=float, default=2e-4, help="learning rate of encoder.")),
    ("--init", dict(type=str, default="xu", choices=["kn", "ku", "xn", "xu"], help="which init scheme to choose.")),
    ("--mode", dict(type=int, default=0, help="Degraded Mode.")),
    ("--natural_scene_single_de_type", dict(nargs="+", default=["gaussianN", "complexN", "blur", "sr", "inpaint", "bandmiss"],
                                            help="which type of single degradation is training and testing for.")),
    ("--remote_sensing_single_de_type", dict(nargs="+", default=["gaussianN", "complexN", "blur", "sr", "inpaint", "haze", "bandmiss"],
                                             help="which type of single degradation is training and testing for.")),
    ("--patch_size", dict(type=int, default=64, help="patchsize of input.")),
    ("--num_workers", dict(type=int, default=16, help="number of workers.")),
    ("--data_type", dict(type=str, default="remote_sensing", help="Types of data used for training.")),
    ("--classifier", dict(type=bool, default=False, help="")),
    ("--db_path", dict(type=str, default="", help="where clean HSIs of remote_sensing saves.")),
    ("--classifier_path", dict(type=str, default="", help="")),
    ("--output_path", dict(type=str, default="output/", help="output save path")),
    ("--ckpt_path", dict(type=str, default=None, help="checkpoint save path")),
    ("--ckpt_dir", dict(type=str, default="", help="Name of the Directory where the checkpoint is to be saved")),
    ("--num_gpus", dict(type=list, default=[0], help="Number of GPUs to use for training")),
    ("--repeat", dict(type=int, default=1, help="")),
    # ---- additions -------------------------------------------------------------------------------
    ("--model", dict(type=str, default=None, choices=[None, "natural_scene", "remote_sensing"],
                     help="which MP_HSIR_Net to build (the reference edits train.py:44-45 by hand); default: --data_type")),
    ("--precision", dict(type=str, default="bf16", choices=["bf16", "f32", "f16"], help="compute dtype of the HIP kernels "
                         "(f16 = the reference's 16-mixed: fp16 compute + dynamic loss scaling)")),
    ("--allow_surrogate_clip", dict(type=int, default=0, help="1: build the model on a seeded stand-in for the CLIP text "
                                    "embeddings when the OpenAI clip package is missing (benchmarks only)")),
    ("--all_sources", dict(type=int, default=0, help="1: do not filter remote-sensing records by source file name")),
    ("--steps_per_epoch", dict(type=int, default=100, help="synthetic source: optimisation steps per epoch")),
    ("--synthetic", dict(type=int, default=1, help="1: GPU-side synthetic patch source (no datasets offline)")),
    ("--log_every", dict(type=int, default=10)),
    ("--graph", dict(type=int, default=1, help="1: capture the training step in a hipGraph after two eager steps and replay it")),
    ("--resume", dict(type=int, default=0, help="1: --ckpt_path also restores the optimizer state and continues at the saved "
                      "epoch + 1 (a checkpoint written by this script); 0 (default) = the reference's warm start: weights only, epoch 0")),
]


def build_parser():
    p = argparse.ArgumentParser()
    for name, kw in _FLAGS:
        p.add_argument(name, **kw)
    return p


options = build_parser().parse_known_args()[0]
