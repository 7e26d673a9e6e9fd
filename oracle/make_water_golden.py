#!/usr/bin/env python3
"""Extracts the golden DATA of the reference's own run logs into tests/golden/*.json:
  /root/reference/water_ccpvdz_out.log        -> water_ccpvdz_rifit.json   (SURVEY 8c golden #1)
  /root/reference/test/water_new_algo-4-8.log -> water_631g2dfp_jkfit.json (golden #2: sp shells, f and g functions)
  /root/reference/test/s10_new_algo-3-20.log  -> s22_10_benzene_methane_631g2dfp_jkfit.json (golden #3: the third run
      in that log, S22 complex 10 benzene...methane, 17 atoms with carbon, 297 AO / 1022 aux, 3 MPI ranks; the log stops
      after the second printed iteration, so the trail has two lines and there is no final energy)
basis + auxiliary basis exponents/coefficients as printed, the COM-shifted geometry in bohr, SCF
settings, the printed iteration trail (iter, E, dE, Drms) and the final energy.
Numbers only — no reference source text is copied.  Run in the build container
(the reference is not present on the GPU box)."""
import json
import os
import re
import sys

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
CASES = [("/root/reference/water_ccpvdz_out.log", "water_ccpvdz_rifit.json",
          "water_ccpvdz_out.log (JuliaChem.jl reference run; lines 44-157, 196-198, 205-216, 253-523, 563)"),
         ("/root/reference/test/water_new_algo-4-8.log", "water_631g2dfp_jkfit.json",
          "test/water_new_algo-4-8.log (JuliaChem.jl reference run; lines 48-202, 226-228, 244-255, 264-283)"),
         ("/root/reference/test/s10_new_algo-3-20.log", "s22_10_benzene_methane_631g2dfp_jkfit.json",
          "test/s10_new_algo-3-20.log (JuliaChem.jl reference run 3 of the file; lines 638-1576, 1587-1603, 1616-1633)", 599)]
AM = {"S": 0, "P": 1, "D": 2, "F": 3, "G": 4}


def parse_basis(lines):
    atoms, cur_atom, cur_shell, last_id = [], None, None, None
    for ln in lines:
        m = re.match(r"Atom #(\d+) \((\w+)\):", ln)
        if m:
            cur_atom = {"symbol": m.group(2), "shells": []}
            atoms.append(cur_atom)
            last_id = None
            continue
        m = re.match(r"\s+(\d+)\s+([SPDFG]|L \([sp]\))\s+(\d+)\s+([-\d.]+)\s+([-\d.]+)\s*$", ln)
        if m and cur_atom is not None:
            # an "L" (sp) shell is printed as its s part followed by its p part under one shell number: two shells
            kind = m.group(2)
            sid = (int(m.group(1)), kind)
            if sid != last_id:
                cur_shell = {"l": AM[kind[3].upper() if kind.startswith("L") else kind], "exps": [], "coefs": []}
                cur_atom["shells"].append(cur_shell)
                last_id = sid
            cur_shell["exps"].append(float(m.group(4)))
            cur_shell["coefs"].append(float(m.group(5)))
    return atoms


def extract(LOG, OUT, source, first_line=0):
    txt = open(LOG).read().splitlines()[first_line:]
    i_aux = next(i for i, l in enumerate(txt) if "Printing Auxillary basis set" in l)
    i_meta = next(i for i, l in enumerate(txt) if "Printing basis set metadata" in l)
    i_bas = next(i for i, l in enumerate(txt) if "Printing basis set..." in l)
    prim = parse_basis(txt[i_bas:i_aux])
    aux = parse_basis(txt[i_aux:i_meta])
    i_xyz = next(i for i, l in enumerate(txt) if "in xyz format" in l)
    geom = []
    i_end = next(i for i, l in enumerate(txt) if "END COORDINATE ANALYSIS" in l)
    for l in txt[i_xyz:i_end]:
        m = re.match(r"^([A-Z][a-z]?)\s+([-\d.eE]+)\s+([-\d.eE]+)\s+([-\d.eE]+)\s*$", l)
        if m:
            geom.append({"symbol": m.group(1), "center": [float(m.group(k)) for k in (2, 3, 4)]})
    trail = []
    for l in txt:
        m = re.match(r"^(\d+)\s+(-?\d+\.\d{10})\s+(-?\d+\.\d{10})\s+(-?\d+\.\d{10})(\s+\d+\.\d{10})?\s*$", l)
        if m:
            trail.append([int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4))])
    e_final = next((float(re.search(r"Total SCF Energy: (-?[\d.]+) h", l).group(1)) for l in txt if "Total SCF Energy" in l), None)
    meta = {}
    for l in txt:
        for key in ("Number of basis functions", "Number of auxillary basis functions", "Number of electrons",
                    "Energy Convergence", "Density Convergence", "DF Max Iterations", "Contraction Mode", "Guess"):
            m = re.match(r"^%s: (.+)$" % re.escape(key), l.strip())
            if m and key not in meta:
                meta[key] = m.group(1)
    out = {"source": source,
           "units": "bohr (COM-shifted, as printed)", "atoms": geom,
           "basis": {a["symbol"]: a["shells"] for a in prim}, "aux_basis": {a["symbol"]: a["shells"] for a in aux},
           "atom_order": [a["symbol"] for a in prim], "charges": {k: v for k, v in {"H": 1, "C": 6, "O": 8}.items() if any(a["symbol"] == k for a in prim)},
           "settings": meta, "trail": trail, "final_energy": e_final}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", os.path.normpath(OUT), len(trail), "iterations, E =", e_final,
          "| shells", [len(a["shells"]) for a in prim], [len(a["shells"]) for a in aux])


def main():
    for log, name, source, *first in CASES:
        extract(log, os.path.join(GOLDEN, name), source, *first)


if __name__ == "__main__":
    sys.exit(main())
