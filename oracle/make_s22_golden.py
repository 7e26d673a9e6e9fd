#!/usr/bin/env python3
"""Extracts the golden DATA the reference's own S22 test runs on (test/runtests.jl:26-63) into tests/golden/s22_cho.json:
  /root/reference/example_inputs/S22/NN_MP2.json   -> symbols + geometry (Angstrom, as in the input file) and charge
  /root/reference/test/s22_gamess_values.json      -> "Energy" (conventional RHF / 6-31G(2df,p), GAMESS), the value
      runtests.jl:62 compares the RHF energy with and :63 allows the density-fitted energy 1.5 mEh around
for the ten complexes made of C, H and O only (2, 3, 8, 9, 10, 11, 16, 17, 20, 22) — the elements the reference's logs
hold 6-31G(2df,p) / cc-pVTZ-JKFIT tables for (tests/golden/water_631g2dfp_jkfit.json: O, H;
tests/golden/s22_10_benzene_methane_631g2dfp_jkfit.json: C, H).  Numbers only — no reference source text is copied.
Run in the build container (the reference is not present on the GPU box)."""
import json
import os

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "s22_cho.json")
NAMES = {2: "water dimer", 3: "formic acid dimer", 8: "methane dimer", 9: "ethene dimer", 10: "benzene - methane",
         11: "benzene dimer (parallel displaced)", 16: "ethene - ethyne", 17: "benzene - water", 20: "benzene dimer (T-shaped)",
         22: "phenol dimer"}


def main():
    gam = json.load(open(os.path.join(REF, "test", "s22_gamess_values.json")))
    out = {"source": "example_inputs/S22/NN_MP2.json (molecule.symbols, molecule.geometry in Angstrom, molecule.molecular_charge) and "
                     "test/s22_gamess_values.json (Energy) of JuliaChem.jl; assertion test/runtests.jl:62-63",
           "angstrom_to_bohr": 1.0 / 0.52917724924,          # JCBasis.jl:61
           "df_tolerance_hartree": 0.0015,                   # runtests.jl:63
           "basis": "6-31G(2df,p)", "auxiliary_basis": "cc-pVTZ-JKFIT", "complexes": {}}
    for k, name in NAMES.items():
        inp = json.load(open(os.path.join(REF, "example_inputs", "S22", "%02d_MP2.json" % k)))
        mol = inp["molecule"]
        assert set(mol["symbols"]) <= {"C", "H", "O"}, (k, mol["symbols"])
        assert inp["model"]["basis"] == "6-31G(2df,p)"
        out["complexes"][str(k)] = {"name": name, "symbols": mol["symbols"], "geometry_angstrom": mol["geometry"],
                                    "molecular_charge": mol.get("molecular_charge", 0),
                                    "gamess_rhf_energy": gam[str(k)]["Energy"]}
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, "with", len(out["complexes"]), "complexes")


if __name__ == "__main__":
    main()
