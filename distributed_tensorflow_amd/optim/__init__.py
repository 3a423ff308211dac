"""Optimizers: fused flat-arena Adam / momentum SGD, and the reference's synchronisation wrappers
by their original class names."""
from .flat import FlatAdam, FlatMomentumSGD  # noqa: F401
from .sync_replicas import (  # noqa: F401
    BackupOptimizer,
    ChiefQueueRunner,
    SoftKillOptimizer,
    SyncReplicasOptimizer,
    SyncReplicasOptimizerV2,
    TimeoutReplicasOptimizer,
    WeightedGradsOptimizer,
)
