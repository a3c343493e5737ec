"""world_size-2 worker (gloo on CPU, oracle-backed engine): the sharded drop-in driver on the plumbing fixture."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import plumbing_util as pu  # noqa: E402
from memvul_amd import model_memory, predict_memory  # noqa: E402


def main():
    root, arch, golden, test_path, out = sys.argv[1:6]
    backend = sys.argv[6] if len(sys.argv) > 6 else "gloo"
    model_memory.Engine = pu.OracleEngine  # tests only: no GPU in this process
    metrics = predict_memory.test_siamese_sharded(
        archive_file=arch, input_file=test_path, input_golden_file=golden, test_config=pu.TEST_CONFIG,
        output_file=os.path.join(root, "test_results", "sharded_metric.json"),
        predictions_output_file=os.path.join(root, "test_results", "sharded_result.json"), batch_size=16,
        engine_options=dict(max_tokens=16 * 256, max_batch=16, max_anchors=16), backend=backend)
    m = predict_memory._jsonable(metrics)
    m["_device_index"] = int(pu.OracleEngine.last_device)  # which GPU this rank's engine was created on
    json.dump(m, open(f"{out}.rank{os.environ.get('RANK', '0')}", "w"))
    import torch.distributed as dist

    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
