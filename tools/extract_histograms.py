"""Extract the atom-count histograms (data, not code) from the reference's datasets_config.py into
jodo_amd/data/n_nodes_hist.json.  Run once in the build container; the JSON is committed.
Source: /root/reference/datasets/datasets_config.py:5-7 (qm9_with_h), :23-25 (qm9_second_half),
:43-57 (geom_with_h_1)."""
import importlib.util
import json
import os

spec = importlib.util.spec_from_file_location("dc", "/root/reference/datasets/datasets_config.py")
dc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(dc)
out = {}
for name in ("qm9_with_h", "qm9_second_half", "geom_with_h_1"):
    info = getattr(dc, name)
    out[name] = {"max_n_nodes": info["max_n_nodes"],
                 "train_n_nodes": {str(k): v for k, v in info["train_n_nodes"].items()}}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "jodo_amd", "data", "n_nodes_hist.json")
with open(dst, "w") as f:
    json.dump(out, f, separators=(",", ":"))
print("wrote", dst)
