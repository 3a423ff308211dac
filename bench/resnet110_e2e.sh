set -e
cd $GRAFT_REPO_ROOT
D=$(mktemp -d)
timeout -k 10 250 python DistributedResNet/src/resnet_distributed_train.py --train_dir $D/train --batch_size 128 --max_steps 30 --log_every 10 --synthetic_data 2>&1 | grep -E "step [0-9]+, loss|Elapsed|Error|Traceback" | tail -5
ls $D/train | head
timeout -k 10 250 python DistributedResNet/src/resnet_eval.py --synthetic_data --checkpoint_dir $D/train --eval_dir $D/eval --run_once=true 2>&1 | grep -E "Precision|loaded|Error|Traceback" | tail -3
python - <<PY
import sys; sys.path.insert(0, ".")
from distributed_tensorflow_amd.checkpoint.saver import latest_checkpoint, bundle_names
p = latest_checkpoint("$D/train"); n = bundle_names(p); print(p.split("/")[-1], len(n), n[:4], n[-3:])
PY
rm -rf $D
