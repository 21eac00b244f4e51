"""Host-side mirror of the reference train loop (microwakeword/train.py) driving the MI355X engine.

``train(model, config, data_processor)`` keeps the reference's per-step behaviour:
  * phase schedule: lists padded with their last entry, piecewise constant by cumulative
    ``training_steps``                                             train.py:168-204,249-263
  * ``model.optimizer.learning_rate.assign(lr)`` every step        train.py:265
  * ``get_data("training", batch_size, spectrogram_length, "default", policy)``   train.py:276-286
  * class weight x penalty weight                                   train.py:288-293 (the reference's arithmetic: SURVEY §A.5, model.combine_weights)
  * ``train_on_batch``                                              train.py:295-299
  * every ``eval_step_interval`` steps: save last weights, ``validate_nonstreaming``, reset metrics,
    best-weights rule, checkpoint                                   train.py:315-451
and ``validate_nonstreaming`` reproduces train.py:41-163 (validation set + 100 ms-stride split of
the ambient set accumulated into the same counters, FAPH / recall-at-cutoff / average viable recall).

When both objects come from this package the spectrogram batch never leaves HBM
(``FeatureHandler.next_training_batch_on_device`` + ``Model.train_on_device_batch``); with any other
duck-typed pair it falls back to the reference's host-array calls.  TensorBoard summaries
(train.py:236-241,328-334,361-389) are written as JSON lines under ``<summaries_dir>``.

Data-parallel (the reference has no distributed code, this wraps its loop train.py:249-299,315-458): when the
process is one rank of an initialised ``torch.distributed`` job (``python -m torch.distributed.run --nproc-per-node N -m
microwakeword_amd.model_train_eval ...``), ``train`` shards every provider's training samples over the ranks, joins the
engines into one gradient exchange (``Model.join_data_parallel``), starts every rank from rank 0's weights and optimizer
state, and runs ``batch_size // W`` windows per rank and step - ``batch_size`` stays the GLOBAL batch of the YAML config.
Validation is sharded by window index and the raw metric counters are summed by one all-reduce per result, so every rank
computes the same validation metrics and takes the same best-weights decision; ONLY RANK 0 writes weights, checkpoints,
summaries and log lines.
"""
from __future__ import annotations

import contextlib
import json
import logging
import os
import sys

import numpy as np

from .model import DEFAULT_WEIGHT_BROADCAST, combine_weights

log = logging.getLogger("microwakeword_amd.train")


@contextlib.contextmanager
def swap_attribute(obj, attr, temp_value):
    original_value = getattr(obj, attr)
    setattr(obj, attr, temp_value)
    try:
        yield
    finally:
        setattr(obj, attr, original_value)


def _trapezoid(y, x):
    fn = getattr(np, "trapezoid", None) or getattr(np, "trapz")
    return fn(y, x)


def _on_device(data_processor, model):
    return hasattr(data_processor, "evaluate_on_device") and hasattr(model, "evaluation_results") \
        and getattr(data_processor, "engine", None) is getattr(model, "engine", object())


def validate_nonstreaming(config, data_processor, model, test_set):
    fast = _on_device(data_processor, model)
    if fast:
        # spectrogram windows never leave HBM
        _, _, result = data_processor.evaluate_on_device(model, test_set, config["spectrogram_length"], "truncate_start", 1024)
    else:
        fingerprints, ground_truth, _ = data_processor.get_data(
            test_set, batch_size=config["batch_size"], features_length=config["spectrogram_length"],
            truncation_strategy="truncate_start")
        ground_truth = ground_truth.reshape(-1, 1)
        model.reset_metrics()
        result = model.evaluate(fingerprints, ground_truth, batch_size=1024, return_dict=True, verbose=0)
    metrics = {k: result[k] for k in ("accuracy", "recall", "precision", "auc", "loss")}
    metrics.update(recall_at_no_faph=0, cutoff_for_no_faph=0, ambient_false_positives=0,
                   ambient_false_positives_per_hour=0, average_viable_recall=0)
    test_set_fp = result["fp"].numpy()

    if data_processor.get_mode_size("validation_ambient") > 0:
        # keep accumulating into the same counters (the reference swaps reset_metrics for a no-op)
        with swap_attribute(model, "reset_metrics", lambda: None):
            if fast:
                _, _, amb = data_processor.evaluate_on_device(model, test_set + "_ambient", config["spectrogram_length"], "split", 1024)
            else:
                amb_x, amb_y, _ = data_processor.get_data(
                    test_set + "_ambient", batch_size=config["batch_size"], features_length=config["spectrogram_length"],
                    truncation_strategy="split")
                amb_y = amb_y.reshape(-1, 1)
                amb = model.evaluate(amb_x, amb_y, batch_size=1024, return_dict=True, verbose=0)
        hours = data_processor.get_mode_duration("validation_ambient") / 3600.0
        all_tp = amb["tp"].numpy()
        ambient_fp = amb["fp"].numpy() - test_set_fp
        all_fn = amb["fn"].numpy()
        metrics["auc"] = amb["auc"]
        metrics["loss"] = amb["loss"]
        with np.errstate(divide="ignore", invalid="ignore"):
            recall_at_cutoffs = all_tp / (all_tp + all_fn)
        faph_at_cutoffs = ambient_fp / hours

        target_cutoff = 1.0
        recall_at_no_faph = 0
        for index, cutoff in enumerate(np.linspace(0.0, 1.0, 101)):
            if faph_at_cutoffs[index] == 0:
                target_cutoff = cutoff
                recall_at_no_faph = recall_at_cutoffs[index]
                break

        if faph_at_cutoffs[0] > 2:
            first = 1
            while faph_at_cutoffs[first] > 2:
                first += 1
            x0, y0 = faph_at_cutoffs[first - 1], recall_at_cutoffs[first - 1]
            x1, y1 = faph_at_cutoffs[first], recall_at_cutoffs[first]
            recall_at_2faph = (y0 * (x1 - 2.0) + y1 * (2.0 - x0)) / (x1 - x0)
        else:
            first = 0
            recall_at_2faph = recall_at_cutoffs[0]
        xs, ys = [2.0], [recall_at_2faph]
        for index in range(first, len(recall_at_cutoffs)):
            if faph_at_cutoffs[index] != xs[-1]:
                xs.append(faph_at_cutoffs[index])
                ys.append(recall_at_cutoffs[index])
        metrics["recall_at_no_faph"] = recall_at_no_faph
        metrics["cutoff_for_no_faph"] = target_cutoff
        metrics["ambient_false_positives"] = ambient_fp[50]
        metrics["ambient_false_positives_per_hour"] = faph_at_cutoffs[50]
        metrics["average_viable_recall"] = _trapezoid(np.flip(ys), np.flip(xs)) / 2.0
    return metrics


def _phase_lists(config):
    def get(key, default):
        v = config.get(key)
        return list(v) if v else list(default)

    lists = dict(
        training_steps=get("training_steps", [20000]), learning_rates=get("learning_rates", [0.001]),
        mix_up_prob=get("mix_up_augmentation_prob", [0.0]), freq_mix_prob=get("freq_mix_augmentation_prob", [0.0]),
        time_mask_max_size=get("time_mask_max_size", [5]), time_mask_count=get("time_mask_count", [2]),
        freq_mask_max_size=get("freq_mask_max_size", [5]), freq_mask_count=get("freq_mask_count", [2]),
        positive_class_weight=get("positive_class_weight", [1.0]), negative_class_weight=get("negative_class_weight", [1.0]))
    n = len(lists["training_steps"])
    for k, v in lists.items():
        while len(v) < n:
            v.append(v[-1])
    return lists


def _penalties_vary(data_processor):
    pens = {float(getattr(p, "penalty_weight", 1.0)) for p in getattr(data_processor, "feature_providers", [])}
    return len(pens) > 1


def _is_better(cur_min, cur_max, best_min, best_max, target):
    return ((cur_min <= target and (cur_max > best_max or best_min > target))
            or (cur_min > target and cur_min < best_min)
            or (cur_min == best_min and cur_max > best_max))


class _JsonSummary:
    def __init__(self, directory, name):
        os.makedirs(directory, exist_ok=True)
        self.f = open(os.path.join(directory, name + ".jsonl"), "a")

    def scalars(self, step, **kv):
        self.f.write(json.dumps(dict(step=int(step), **{k: float(v) for k, v in kv.items()})) + "\n")
        self.f.flush()


def process_group():
    """(rank, world) of the ``torch.distributed`` job this process is a rank of; (0, 1) outside one.  torch is not imported
    for the question: a process that never imported ``torch.distributed`` cannot have initialised a group."""
    dist = sys.modules.get("torch.distributed")
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class _NoSummary:
    def scalars(self, step, **kv):
        pass


def train(model, config, data_processor, verbose=True):
    ph = _phase_lists(config)
    model.compile()
    model.make_train_function()
    rank, world = process_group()
    chief = rank == 0
    fast = hasattr(data_processor, "next_training_batch_on_device") and hasattr(model, "train_on_device_batch") \
        and getattr(data_processor, "engine", None) is getattr(model, "engine", object())
    prefetch = int(config.get("prefetch_batches", 4))
    ckpt_dir = os.path.join(config["train_dir"], "restore")
    ckpt = os.path.join(ckpt_dir, "ckpt")
    if chief and os.path.isfile(ckpt + ".weights.npz") and hasattr(model, "load_optimizer_state"):
        # restore is unconditional in the reference (train.py:232-233); data-parallel: rank 0 restores, the broadcast below
        # hands weights, moving statistics and Adam state to the other ranks
        model.load_weights(ckpt + ".weights")
        model.load_optimizer_state(ckpt + ".opt.npz")

    dp = None
    if world > 1 or config.get("data_parallel"):
        if not fast:
            raise ValueError("data-parallel training needs this package's Model and FeatureHandler on one engine "
                             "(the spectrogram batches are assembled per rank in HBM)")
        if config["batch_size"] % world:
            raise ValueError("batch_size %d (the global batch) is not divisible by the %d ranks" % (config["batch_size"], world))
        from .parallel import shard_feature_handler
        dp = model.data_parallel or model.join_data_parallel(sync_bn=bool(config.get("sync_bn", False)),
                                                             grad_buckets=int(config.get("grad_buckets", 1)))
        dp.broadcast_parameters(0, optimizer_state=True)
        # rank-distinct sampler streams, derived from the configured seed (or from the global stream as it stands) and from
        # the optimizer step the run (re)starts at: a relaunch continues with fresh draws
        resumed_step = int(model.engine.get_opt_state()[2]) if hasattr(model.engine, "get_opt_state") else 0
        seed = config.get("data_parallel_seed")
        shard_feature_handler(data_processor, rank, world, seed=None if seed is None else int(seed), prefetch=prefetch, epoch=resumed_step)
        log.info("data-parallel: rank %d of %d, %d windows per rank and step (global batch %d), %s BatchNorm", rank, world,
                 config["batch_size"] // world, config["batch_size"], "synchronised" if dp.sync_bn else "rank-local")
    elif fast and hasattr(data_processor, "use_private_rng") and prefetch > 0:
        # the draws of get_data("training") (data.py:540-569) continue the global random / numpy.random streams from where
        # they stand now, on a worker thread that runs `prefetch_batches` batches ahead of the step being enqueued
        # (native.Prefetcher).  The streams are handed back around every validation pass (below), whose shuffles advance
        # the global numpy stream in the reference too.  prefetch_batches: 0 keeps every draw on this thread and in the
        # global generators, as the reference does.
        data_processor.use_private_rng(prefetch=prefetch)
    private_streams = fast and hasattr(data_processor, "release_private_rng") and (dp is not None or prefetch > 0)
    local_batch = config["batch_size"] // world
    # The reference prints the running metrics after every step (train.py:302-312), which costs a host-device synchronisation
    # per step (0.50 instead of 0.31 ms per step at batch 1024).  On the device path the numbers are read back every
    # `progress_interval_steps` steps (and at every evaluation boundary); the counters themselves accumulate on the device
    # every step, so what is printed is what the reference would print at that step.  1 = the reference's cadence.
    progress = max(1, int(config.get("progress_interval_steps", 25)))

    train_writer = _JsonSummary(os.path.join(config["summaries_dir"], "train"), "scalars") if chief else _NoSummary()
    val_writer = _JsonSummary(os.path.join(config["summaries_dir"], "validation"), "scalars") if chief else _NoSummary()

    warned_weights = False
    steps_max = int(np.sum(ph["training_steps"]))
    best_min, best_max, best_cutoff = 10000, 0.0, 1.0

    def save_weights(path):
        if chief:
            model.save_weights(path)

    def save_ckpt():
        if not chief:
            return
        os.makedirs(ckpt_dir, exist_ok=True)
        model.save_weights(ckpt + ".weights")
        if hasattr(model, "save_optimizer_state"):
            model.save_optimizer_state(ckpt + ".opt.npz")

    for step in range(1, steps_max + 1):
        acc = 0
        for i, n in enumerate(ph["training_steps"]):
            acc += n
            if step <= acc:
                break
        lr = ph["learning_rates"][i]
        model.optimizer.learning_rate.assign(lr)
        policy = {"mix_up_prob": ph["mix_up_prob"][i], "freq_mix_prob": ph["freq_mix_prob"][i],
                  "time_mask_max_size": ph["time_mask_max_size"][i], "time_mask_count": ph["time_mask_count"][i],
                  "freq_mask_max_size": ph["freq_mask_max_size"][i], "freq_mask_count": ph["freq_mask_count"][i]}
        cw_neg, cw_pos = ph["negative_class_weight"][i], ph["positive_class_weight"][i]
        mode = config.get("sample_weight_broadcast", DEFAULT_WEIGHT_BROADCAST)
        if fast:
            data_processor.next_training_batch_on_device(local_batch, config["spectrogram_length"], "default", policy,
                                                         class_weights=(cw_neg, cw_pos), weight_broadcast=mode)
            boundary = (step % config["eval_step_interval"]) == 0 or step == steps_max
            want = boundary or (verbose and chief and (step % progress == 0 or step == 1))
            result = model.train_on_device_batch(local_batch, want_results=want)
        else:
            x, y, w = data_processor.get_data("training", batch_size=config["batch_size"],
                                              features_length=config["spectrogram_length"], truncation_strategy="default",
                                              augmentation_policy=policy)
            combined = combine_weights(w, y, cw_neg, cw_pos, mode)
            result = model.train_on_batch(x, y.reshape(-1, 1), sample_weight=combined)
        if cw_neg != cw_pos and not warned_weights and _penalties_vary(data_processor):
            # the readings of train.py:288-293 only part ways when class AND penalty weights are non-uniform
            warned_weights = True
            log.warning("class weights %g / %g and the providers' penalty weights are both non-uniform: this run follows %s "
                        "(sample_weight_broadcast: %s).  keras_last_axis = the reference's arithmetic as Keras 3 reduces its [B,B] weight "
                        "matrix (train.py:288-293): mean_j(penalty_j x bce_j) x mean_i(class_weight(y_i)) - the class weights only rescale "
                        "the batch loss; per_sample = the evident intent, mean_i(penalty_i x class_weight(y_i) x bce_i).  INTEGRATION.md "
                        "section 2.", cw_neg, cw_pos, "the reference's arithmetic" if mode == "keras_last_axis" else "this reading", mode)
        if verbose and chief and result is not None:
            print("Validation Batch #{:d}: Accuracy = {:.3f}; Recall = {:.3f}; Precision = {:.3f}; Loss = {:.4f}; Mini-Batch #{:d}".format(
                (step // config["eval_step_interval"] + 1), result[1], result[2], result[3], result[9],
                (step % config["eval_step_interval"])), end="\r")

        is_last = step == steps_max
        if (step % config["eval_step_interval"]) == 0 or is_last:
            if dp is not None:
                dp.average_bn_state()   # rank-local BatchNorm: validate and save ONE model (the mean of the ranks' moving statistics)
                # the running train metrics over every rank's windows (each rank's counters cover its own batches)
                tm = model._metric_results(reduce=True)
                result = [result[0], tm["accuracy"], tm["recall"], tm["precision"], tm["tp"], tm["fp"], tm["tn"], tm["fn"], tm["auc"], tm["loss"]]
            info = log.info if chief else log.debug
            info("Step #%d: rate %f, accuracy %.2f%%, recall %.2f%%, precision %.2f%%, cross entropy %f",
                 step, lr, result[1] * 100, result[2] * 100, result[3] * 100, result[9])
            train_writer.scalars(step, loss=result[9], accuracy=result[1], recall=result[2], precision=result[3], auc=result[8])
            save_weights(os.path.join(config["train_dir"], "last_weights.weights.h5"))
            if private_streams:
                data_processor.release_private_rng()   # validation shuffles on the global numpy stream (data.py:593-595) ...
            nm = validate_nonstreaming(config, data_processor, model, "validation")
            if private_streams:
                data_processor.use_private_rng(prefetch=prefetch)   # ... and the training draws continue behind them
            model.reset_metrics()
            info("Step %d (nonstreaming): Validation: recall at no faph = %.3f with cutoff %.2f, accuracy = %.2f%%, recall = %.2f%%, "
                     "precision = %.2f%%, ambient false positives = %d, estimated false positives per hour = %.5f, loss = %.5f, "
                     "auc = %.5f, average viable recall = %.9f", step, nm["recall_at_no_faph"] * 100, nm["cutoff_for_no_faph"],
                 nm["accuracy"] * 100, nm["recall"] * 100, nm["precision"] * 100, nm["ambient_false_positives"],
                 nm["ambient_false_positives_per_hour"], nm["loss"], nm["auc"], nm["average_viable_recall"])
            val_writer.scalars(step, loss=nm["loss"], accuracy=nm["accuracy"], recall=nm["recall"], precision=nm["precision"],
                               recall_at_no_faph=nm["recall_at_no_faph"], auc=nm["auc"], average_viable_recall=nm["average_viable_recall"])
            if chief:
                os.makedirs(os.path.join(config["train_dir"], "train"), exist_ok=True)
            save_weights(os.path.join(config["train_dir"], "train", f"{int(best_min * 10000)}_weights_{step}.weights.h5"))
            cur_min = 0.0 if config["minimization_metric"] is None else nm[config["minimization_metric"]]
            cur_max = nm[config["maximization_metric"]]
            if _is_better(cur_min, cur_max, best_min, best_max, config["target_minimization"]):
                best_min, best_max, best_cutoff = cur_min, cur_max, nm["cutoff_for_no_faph"]
                save_weights(os.path.join(config["train_dir"], "best_weights.weights.h5"))
                save_ckpt()
            info("So far the best minimization quantity is %.3f with best maximization quantity of %.5f%%; no faph cutoff is %.2f",
                 best_min, best_max * 100, best_cutoff)
    save_ckpt()
    save_weights(os.path.join(config["train_dir"], "last_weights.weights.h5"))
    if dp is not None:
        dp.barrier()   # rank 0's files are complete before any rank returns (and tears the process group down)
    return dict(best_minimization=best_min, best_maximization=best_max, best_no_faph_cutoff=best_cutoff)
