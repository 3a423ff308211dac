"""Does hipEventQuery fail for an event that was recorded on a stream BEFORE that stream started
capturing, when the query comes from another thread during the capture? (The RCCL process group's
watchdog thread polls the end events of collectives this way.)

    python event_query_during_capture.py same    capture on the event's stream: the query raises
                                                 "operation not permitted when stream is capturing"
                                                 and the capture is invalidated (measured, ROCm 7.2)
    python event_query_during_capture.py other   capture on another stream: clean

Hence Program::instantiate (csrc/runtime/program.cc) captures on a stream of its own."""
import sys
import threading
import torch

MODE = sys.argv[1] if len(sys.argv) > 1 else "other"

s = torch.cuda.Stream()
cap = s if MODE == "same" else torch.cuda.Stream()
e = torch.cuda.Event()
x = torch.ones(1024, device="cuda")
with torch.cuda.stream(s):
    x += 1
    e.record(s)
s.synchronize()
stop, errors, polls = False, [], 0


def poll():
    global polls
    while not stop:
        try:
            e.query()
            polls += 1
        except Exception as ex:  # noqa: BLE001
            errors.append(str(ex)[:200])
            return


t = threading.Thread(target=poll)
t.start()
for i in range(300):
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
            x += 1
    except Exception as ex:  # noqa: BLE001
        errors.append("capture: " + str(ex)[:160])
        break
stop = True
t.join()
print(f"mode {MODE}: polls {polls}, errors {len(errors)}", errors[:2])
