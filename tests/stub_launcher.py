"""Stand-in for `python -m torch.distributed.run` in tests/test_bench_launcher.py: prints its own argument list as one JSON line."""
import json
import os
import sys

print("launcher chatter that is not the result line")
print(json.dumps({"stub_argv": sys.argv[1:], "ipc": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}))
sys.exit(int(os.environ.get("STUB_EXIT", "0")))
