"""Learning-rate schedule of the reference training drivers.

lr = initial_learning_rate * decay_factor ^ floor(global_step / decay_steps)   (staircase)
decay_steps = int(num_batches_per_epoch * num_epochs_per_decay / num_replicas_to_aggregate)
(reference: DistributedLeNet/src/distributed_train.py:139-152,
 DistributedResNet/src/distributed_train.py:213-234)
"""
from __future__ import annotations


def decay_steps(num_examples: int, batch_size: int, num_epochs_per_decay: float, num_replicas_to_aggregate: int) -> int:
    num_batches_per_epoch = num_examples / batch_size
    return max(1, int(num_batches_per_epoch * num_epochs_per_decay / num_replicas_to_aggregate))


def exponential_decay(initial_lr: float, global_step: int, steps: int, decay_factor: float, staircase: bool = True) -> float:
    p = global_step / steps
    if staircase:
        p = global_step // steps
    return initial_lr * (decay_factor ** p)


class StaircaseSchedule:
    """The drivers' schedule as a function of the global step, with its three numbers exposed so
    that the device-side aggregation kernel can evaluate the same staircase for any update it
    applies (a replica that catches up applies several updates, each with its own rate)."""

    def __init__(self, initial_lr: float, steps: int, decay_factor: float):
        self.initial_lr, self.steps, self.decay_factor = float(initial_lr), max(1, int(steps)), float(decay_factor)

    def __call__(self, global_step: int) -> float:
        return exponential_decay(self.initial_lr, int(global_step), self.steps, self.decay_factor)


def schedule_params(lr) -> tuple:
    """(initial_lr, decay_steps, decay_factor) of a StaircaseSchedule or of a constant rate."""
    if isinstance(lr, StaircaseSchedule):
        return lr.initial_lr, lr.steps, lr.decay_factor
    if callable(lr):
        raise TypeError("a device-side update log needs a StaircaseSchedule or a constant learning rate")
    return float(lr), 1, 1.0


def piecewise_resnet(init_lr: float, step: int, decay_step0: int, decay_step1: int, factor: float) -> float:
    """Single-machine schedule of the ResNet hyper-parameter file (decay at two fixed steps;
    reference: DistributedResNet/src/hyper_parameters.py:33-38)."""
    lr = init_lr
    if step >= decay_step0:
        lr *= factor
    if step >= decay_step1:
        lr *= factor
    return lr
