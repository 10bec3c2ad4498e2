"""Collect the parity reports the GPU tests write (gpurun_out/parity_*.json, tests/test_model_gpu.py::_oracle_vs_hip_step)
into one tracked file: python tools/make_parity_profile.py r3  ->  profiles/r3_parity.json"""
import glob, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
out = {"what": "HIP path vs CPU oracle, one whole training step on identical weights / batch / noise draws; errors are maxima over "
               "all tensors of a class, relative as named (tests/test_model_gpu.py::_oracle_vs_hip_step); bar: 1e-4 fp32, 2e-2 bf16",
       "cases": {}}
for path in sorted(glob.glob(os.path.join(root, "gpurun_out", "parity_*.json"))):
    r = json.load(open(path))
    name = os.path.basename(path)[len("parity_"):-len(".json")]
    keep = {k: r[k] for k in ("cfg", "B", "precision", "step", "noise", "losses_hip", "losses_oracle", "loss_rel_err", "grad_norm_hip",
                              "grad_norm_oracle", "grad_norm_oracle_fp64", "grad_norm_rel_err", "clipped", "relu_gates",
                              "grad_err_vs_tensor_max", "weight_err_solid_vs_tensor_max", "weight_err_in_adam_steps") if k in r}
    keep["classes"] = {c: {k: v for k, v in d.items() if k != "weight_err_vs_tensor_max"} for c, d in r["classes"].items()}
    out["cases"][name] = keep
with open(os.path.join(root, "profiles", f"{tag}_parity.json"), "w") as f:
    json.dump(out, f, indent=1)
for n, c in out["cases"].items():
    print(f"{n:34s} loss {c['loss_rel_err']:.1e} norm {c['grad_norm_rel_err']:.1e} grads {c['grad_err_vs_tensor_max']:.1e} "
          f"weights {c['weight_err_solid_vs_tensor_max']:.1e} flips {c.get('relu_gates', {}).get('flipped')}/{c.get('relu_gates', {}).get('compared')}")
