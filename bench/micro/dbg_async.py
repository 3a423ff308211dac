"""Per-step loss of the asynchronous parameter-shard path on one GPU (debug aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from distributed_tensorflow_amd.parallel import control_plane
from distributed_tensorflow_amd.parallel.async_ps import AsyncParameterShards
from distributed_tensorflow_amd.parallel.comm import Communicator
from distributed_tensorflow_amd.train.replica import HipReplica

model = sys.argv[1] if len(sys.argv) > 1 else "resnet"
comm = Communicator.single("cuda")
torch.manual_seed(0)
B = int(os.environ.get("B", "256"))
rep = HipReplica(model, B, num_residual_blocks=3)
name = control_plane.board_name_for("dbg_async", 1)
board = control_plane.open_board(name, 1, 0)
sh = AsyncParameterShards(rep, comm, board, 4, name + "_ps", lambda v: 1e-3)
if model == "resnet":
    x = torch.randn(B, 32, 32, 3, device="cuda")
else:
    x = torch.randn(B, 28, 28, 1, device="cuda")
y = torch.randint(0, 10, (B,), device="cuda")
rep.load_batch(x, y)
for i in range(int(os.environ.get("STEPS", "12"))):
    loss, gs = sh.step()
    p = rep.arena.params
    print(i, loss, gs, float(p.abs().max()), bool(torch.isfinite(p).all()), bool(torch.isfinite(rep.arena.grads).all()), flush=True)
type(board).unlink(name)
