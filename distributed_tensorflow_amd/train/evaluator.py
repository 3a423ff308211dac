"""Continuous evaluator: polls the training directory for new checkpoints and scores them.

Mirrors the reference's evaluator processes (reference: DistributedLeNet/src/mnist_eval.py:30-35,
DistributedLeNet/src/nn_eval.py:66-140, DistributedResNet/src/resnet_eval.py,
DistributedResNet/src/nn_eval.py:127-183): wipe and recreate `eval_dir`, build the model for
inference, then every `eval_interval_secs` read the `checkpoint` state file, restore the newest
bundle, skip it if its step was already evaluated, and print
  `Succesfully loaded model from <path> at step=<step>.`
  `Num examples: %d  Precision @ 1: %f Loss: %f Time: %f`
(text kept verbatim, the benchmark tooling greps for it) and write the `Validation Accuracy` /
`Validation Loss` scalar summaries. Trainer and evaluator are coupled only through the
checkpoint directory. The reference evaluates the whole validation set as ONE batch, which
matters for the ResNet because its BatchNorm uses the statistics of the evaluated batch;
`--eval_batch_size=0` (default) keeps that, a positive value evaluates in chunks.
"""
from __future__ import annotations

import logging
import os
import shutil
import time
from typing import Dict, Optional, Tuple

import numpy as np

from ..checkpoint.saver import Saver, get_checkpoint_state, step_of
from ..data import cifar10 as cifar_data
from ..data import mnist as mnist_data
from .driver import setup_logging
from .replica import make_replica

log = logging.getLogger("dtf")

LOADED_LINE = "Succesfully loaded model from %s at step=%s."
RESULT_LINE = "Num examples: %d  Precision @ 1: %f Loss: %f Time: %f"


def load_validation(model: str, data_dir: str, synthetic: bool) -> Tuple[np.ndarray, np.ndarray]:
    if model == "lenet":
        sets = mnist_data.load_mnist("" if synthetic else data_dir)
        val = sets.validation
        if val.fake_data:
            return val.next_batch(1000)
        return val.images, val.labels
    return cifar_data.read_validation_data("" if synthetic else data_dir)


class Evaluator:
    def __init__(self, flags, model: str):
        self.flags, self.model = flags, model
        self.images, self.labels = load_validation(model, flags.data_dir, flags.synthetic_data)
        n = len(self.labels)
        self.batch = n if flags.eval_batch_size <= 0 else min(flags.eval_batch_size, n)
        self.replica = make_replica(model, self.batch, flags.device, num_residual_blocks=flags.num_residual_blocks,
                                    optimizer=flags.optimizer, use_graph=False, inference_only=True)
        self.saver = Saver(self.replica.arena, flags.optimizer, flags.checkpoint_names)
        self.last_step: Optional[int] = None
        self.writer = None

    def do_eval(self) -> Optional[Dict]:
        st = get_checkpoint_state(self.flags.checkpoint_dir)
        if st is None or not os.path.exists(st[0] + ".index"):
            log.info("No checkpoint file found")
            return None
        prefix = st[0]
        global_step = step_of(prefix)
        if self.last_step is not None and global_step == self.last_step:
            return None                                     # already scored: nothing is read or copied
        try:
            self.saver.restore(prefix, strict=True)
        except (RuntimeError, KeyError, ValueError) as e:   # the trainer may be rotating files
            log.info("Could not load %s yet: %s", prefix, e)
            return None
        self.last_step = global_step
        log.info(LOADED_LINE % (prefix, global_step))
        t0 = time.time()
        n = len(self.labels)
        correct, loss_sum, seen = 0, 0.0, 0
        for lo in range(0, n - self.batch + 1, self.batch):
            l, c = self.replica.evaluate(self.images[lo:lo + self.batch], self.labels[lo:lo + self.batch])
            correct += c
            loss_sum += l * self.batch
            seen += self.batch
        precision = correct / max(1, seen)
        loss = loss_sum / max(1, seen)
        dt = time.time() - t0
        log.info(RESULT_LINE % (seen, precision, loss, dt))
        if self.writer is not None:
            self.writer.add_scalars({"Validation Accuracy": precision, "Validation Loss": loss}, global_step)
        return {"step": global_step, "precision": precision, "loss": loss, "examples": seen, "time": dt}

    def run(self) -> list:
        from ..utils.summary import FileWriter

        f = self.flags
        if os.path.exists(f.eval_dir):
            shutil.rmtree(f.eval_dir)
        os.makedirs(f.eval_dir)
        self.writer = FileWriter(f.eval_dir)
        results = []
        while True:
            r = self.do_eval()
            if r is not None:
                results.append(r)
            if f.run_once or (f.max_evals and len(results) >= f.max_evals):
                break
            time.sleep(f.eval_interval_secs)
        self.writer.close()
        return results


def evaluate(flags, model: str) -> list:
    setup_logging()
    return Evaluator(flags, model).run()
