#!/usr/bin/env python3
"""An input file of the reference's format through the DF-RHF path on the device — the sequence of the reference's
example_scripts/minimal-rhf.jl (JCInput.run -> JCBasis.run -> JCRHF.Energy.run).

  python tools/run_input.py water.json --basis-path /path/to/tables [--output 2] [--device 0] [--set niter=50 ...]

Basis tables: <name>.json | .gbs | .nw files (Basis Set Exchange JSON, Gaussian94, NWChem) named after model["basis"] and
model["auxiliary_basis"], in --basis-path or $JCDF_BASIS_PATH.  Under torchrun every rank runs this script; rank 0 prints."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("input")
    ap.add_argument("--basis-path", action="append", default=[])
    ap.add_argument("--output", type=int, default=2)
    ap.add_argument("--device", type=int, default=None)
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="override a keywords.scf entry")
    a = ap.parse_args()
    import torch
    from juliachem_jl_amd import inputs
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        a.device = local
    over = {"scf_type": "df", "contraction_mode": "GPU"}
    for kv in a.set:
        k, v = kv.split("=", 1)
        try:
            over[k] = json.loads(v)
        except ValueError:
            over[k] = v
    rank = int(os.environ.get("RANK", "0"))
    res = inputs.run_input(a.input, inputs.BasisLibrary(a.basis_path), output=a.output if rank == 0 else 0,
                           device=a.device, scf_overrides=over)
    if rank == 0:
        print("Energy %.10f Eh   converged %s   iterations %d   wall %.2f s"
              % (res["Energy"], res["Converged?"], res["Iterations"], res["Timings"].run_time))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0 if res["Converged?"] else 1


if __name__ == "__main__":
    sys.exit(main())
