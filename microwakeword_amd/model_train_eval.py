"""CLI / config surface of the reference kept for the training path:
``python -m microwakeword_amd.model_train_eval --training_config cfg.yaml mixednet --residual_connection "0,0,0,0"``

Mirrors microwakeword/model_train_eval.py:
  * ``load_config(flags, model_module)``     :45-96   (YAML keys + derived ``summaries_dir, stride,
    spectrogram_length_final_layer, spectrogram_length, flags, training_input_shape``)
  * ``train_model(config, model, data_processor, restore_checkpoint)``   :99-128
  * argparse surface                          :277-389 (the ``--test_*`` export flags are accepted; TFLite
    export / streaming evaluation stay with the reference and raise here if requested)

Data-parallel over the GPUs of one node (no reference equivalent): launched as
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m
microwakeword_amd.model_train_eval --training_config cfg.yaml mixednet ...`` every process reads ``RANK`` / ``LOCAL_RANK`` /
``WORLD_SIZE``, takes GPU ``LOCAL_RANK``, joins an RCCL process group and ``train.train`` does the rest (sharded providers,
gradient all-reduce inside the step, sharded validation, rank 0 writes the files).  ``batch_size`` of the YAML stays the
global batch.
"""
from __future__ import annotations

import argparse
import logging
import os
import sys

import yaml

from . import inception
from . import mixednet
from .data import FeatureHandler
from . import train as train_mod


def get_input_data_shape(config):
    """layers/modes.py:40-64 for the TRAINING / NON_STREAM_INFERENCE modes."""
    return (config["spectrogram_length"], 40)


def load_config(flags, model_module):
    config = yaml.load(open(flags.training_config, "r").read(), yaml.Loader)
    config["summaries_dir"] = os.path.join(config["train_dir"], "logs/")
    config["stride"] = flags.__dict__.get("stride", 1)
    config["window_step_ms"] = config.get("window_step_ms", 20)
    sample_rate, window_size_ms = 16000, 30
    desired_samples = int(sample_rate * config["clip_duration_ms"] / 1000)
    window_size_samples = int(sample_rate * window_size_ms / 1000)
    window_step_samples = int(config["stride"] * sample_rate * config["window_step_ms"] / 1000)
    length_minus_window = desired_samples - window_size_samples
    if length_minus_window < 0:
        config["spectrogram_length_final_layer"] = 0
    else:
        config["spectrogram_length_final_layer"] = 1 + int(length_minus_window / window_step_samples)
    config["spectrogram_length"] = config["spectrogram_length_final_layer"] + model_module.spectrogram_slices_dropped(flags)
    config["flags"] = flags.__dict__
    config["training_input_shape"] = get_input_data_shape(config)
    return config


def save_model_summary(model, path, file_name="model_summary.txt"):
    """utils.py:131-145."""
    with open(os.path.join(path, file_name), "wt") as fd:
        model.summary(print_fn=lambda x: fd.write(x + "\n"))


def claim_train_dir(config, restore_checkpoint):
    """model_train_eval.py:99-120: the run owns a fresh ``train_dir`` unless it restores a checkpoint.  In a data-parallel
    job rank 0 creates the directory and every rank learns the outcome (one object broadcast), so that all of them raise
    - or none."""
    rank, world = train_mod.process_group()
    exists = False
    if rank == 0:
        try:
            os.makedirs(config["train_dir"])
            os.mkdir(config["summaries_dir"])
        except OSError:
            exists = True
    if world > 1:
        import torch.distributed as dist
        box = [exists]
        dist.broadcast_object_list(box, src=0)
        exists = box[0]
    if exists and not restore_checkpoint:
        raise ValueError("model already exists in folder %s" % config["train_dir"]) from None


def train_model(config, model, data_processor, restore_checkpoint):
    """model_train_eval.py:99-128: claims ``train_dir`` (a fresh directory, or an existing one only with
    ``restore_checkpoint``: "model already exists" otherwise), writes the configuration and the model summary, trains.
    Called once the model and the data processor exist - as in the reference - so a set-up failure leaves no directory."""
    claim_train_dir(config, restore_checkpoint)
    if train_mod.process_group()[0] == 0:
        with open(os.path.join(config["train_dir"], "training_config.yaml"), "w") as outfile:
            yaml.dump({k: v for k, v in config.items() if k != "features" or all("stores" not in f for f in v)}, outfile,
                      default_flow_style=False)
        save_model_summary(model, config["train_dir"])
    return train_mod.train(model, config, data_processor)


def init_process_group_from_env():
    """One process per GPU: ``RANK`` / ``LOCAL_RANK`` / ``WORLD_SIZE`` / ``MASTER_ADDR`` / ``MASTER_PORT`` as
    ``torch.distributed.run`` exports them.  Returns (rank, local_rank, world); (0, None, 1) outside such a launch - torch is
    not imported then.  The backend is RCCL (``"nccl"``); ``MWW_DIST_BACKEND=gloo`` serves host-emulated builds of the
    library (tests)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, None, 1
    import torch
    import torch.distributed as dist
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    backend = os.environ.get("MWW_DIST_BACKEND", "nccl")
    if not dist.is_initialized():
        if backend == "nccl":
            # 88 KB of gradient per step: latency-bound, one or two channels move it as fast as many
            os.environ.setdefault("NCCL_MAX_NCHANNELS", "2")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--training_config", type=str, default="trained_models/model/training_parameters.yaml")
    parser.add_argument("--train", type=int, default=1)
    parser.add_argument("--test_tf_nonstreaming", type=int, default=0)
    parser.add_argument("--test_tflite_nonstreaming", type=int, default=0)
    parser.add_argument("--test_tflite_nonstreaming_quantized", type=int, default=0)
    parser.add_argument("--test_tflite_streaming", type=int, default=0)
    parser.add_argument("--test_tflite_streaming_quantized", type=int, default=0)
    parser.add_argument("--restore_checkpoint", type=int, default=0)
    parser.add_argument("--use_weights", type=str, default="best_weights")
    parser.add_argument("--verbosity", type=str, default="INFO")
    parser.add_argument("--device", type=int, default=0, help="HIP device index (one process per GPU)")
    subparsers = parser.add_subparsers(dest="model_name", help="NN model name")
    inception.model_parameters(subparsers.add_parser("inception"))
    mixednet.model_parameters(subparsers.add_parser("mixednet"))
    return parser


def main(argv=None):
    parser = build_parser()
    flags, unparsed = parser.parse_known_args(argv)
    if unparsed:
        raise ValueError("Unknown argument: {}".format(unparsed))
    if flags.model_name == "mixednet":
        model_module = mixednet
    elif flags.model_name == "inception":
        model_module = inception
    else:
        raise ValueError("Unknown model type: {}".format(flags.model_name))
    already = "torch.distributed" in sys.modules and sys.modules["torch.distributed"].is_initialized()
    rank, local_rank, world = init_process_group_from_env()
    logging.basicConfig(level=getattr(logging, flags.verbosity.upper(), logging.INFO) if rank == 0 else logging.WARNING)
    try:
        return _run(flags, model_module, rank, local_rank, world)
    finally:
        if world > 1 and not already:   # the group this call created (a caller's own group is the caller's to end)
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()


def _run(flags, model_module, rank, local_rank, world):
    if any((flags.test_tf_nonstreaming, flags.test_tflite_nonstreaming, flags.test_tflite_nonstreaming_quantized,
            flags.test_tflite_streaming, flags.test_tflite_streaming_quantized)):
        raise NotImplementedError("model export / TFLite evaluation stays with the reference (microwakeword.utils / .test); "
                                  "train here, then load the saved weights there (INTEGRATION.md)")
    config = load_config(flags, model_module)
    if flags.train:
        device = flags.device if local_rank is None else local_rank
        if world > 1 and config["batch_size"] % world:
            raise ValueError("batch_size %d (the global batch) is not divisible by the %d ranks" % (config["batch_size"], world))
        # every rank's engine holds batch_size / W windows per step
        model = model_module.model(flags, config["training_input_shape"], config["batch_size"] // world, device=device)
        from .train import process_group
        rank, world = process_group()
        # a data-parallel rank uploads its shard of the training samples only (train() would shard an unsharded handler too)
        data_processor = FeatureHandler(config, engine=model.engine, shard=(rank, world) if world > 1 else None)
        if rank == 0:
            model.summary(print_fn=logging.getLogger("microwakeword_amd").info)
        return train_model(config, model, data_processor, flags.restore_checkpoint)
    if not os.path.isdir(config["train_dir"]):
        raise ValueError('model is not trained set "--train 1" and retrain it')


if __name__ == "__main__":
    main(sys.argv[1:])
