"""Input files and basis tables (juliachem.jl_amd/inputs.py): the three text formats give back the golden fixtures' tables,
the reference's input JSON and .xyz conventions (JCInput.jl:34-82, xyz_to_molecule.jl:3-29, JCBasis.jl:57-61), and
— on the GPU — an input file run end to end lands on the reference's energy."""
import json
import os

import numpy as np
import pytest

from juliachem_jl_amd import inputs
from water_case import FIXTURES, GOLDEN

LETTER = "SPDFGHI"


def golden(case):
    return json.load(open(os.path.join(GOLDEN, FIXTURES[case])))


def blocks(shells):
    """Group an s shell and a following p shell with the same exponents into one sp block (how the tables are published)."""
    out, i = [], 0
    while i < len(shells):
        a = shells[i]
        if a["l"] == 0 and i + 1 < len(shells) and shells[i + 1]["l"] == 1 and shells[i + 1]["exps"] == a["exps"]:
            out.append(([0, 1], a["exps"], [a["coefs"], shells[i + 1]["coefs"]]))
            i += 2
        else:
            out.append(([a["l"]], a["exps"], [a["coefs"]]))
            i += 1
    return out


def to_gbs(tab):
    t = ["! written by the test", "****"]
    for sym, shells in tab.items():
        t.append("%s     0" % sym)
        for ams, exps, cols in blocks(shells):
            t.append("%s   %d   1.00" % ("".join(LETTER[a] for a in ams), len(exps)))
            for k, e in enumerate(exps):
                t.append(("   %r" % e).replace("e", "D") + "".join("   %r" % c[k] for c in cols))
        t.append("****")
    return "\n".join(t) + "\n"


def to_nw(tab):
    t = ["# written by the test", 'BASIS "ao basis" PRINT']
    for sym, shells in tab.items():
        for ams, exps, cols in blocks(shells):
            t.append("%s    %s" % (sym, "".join(LETTER[a] for a in ams)))
            for k, e in enumerate(exps):
                t.append("   %r" % e + "".join("   %r" % c[k] for c in cols))
    t.append("END")
    return "\n".join(t) + "\n"


def to_bse(tab):
    els = {}
    for sym, shells in tab.items():
        els[str(inputs.ATOMIC_NUMBER[sym])] = {"electron_shells": [
            {"function_type": "gto", "angular_momentum": ams, "exponents": [repr(e) for e in exps],
             "coefficients": [[repr(x) for x in c] for c in cols]} for ams, exps, cols in blocks(shells)]}
    return json.dumps({"molssi_bse_schema": {"schema_type": "complete", "schema_version": "0.1"}, "elements": els})


@pytest.mark.parametrize("case", ["ccpvdz", "631g2dfp"])
@pytest.mark.parametrize("writer", [to_gbs, to_nw, to_bse])
@pytest.mark.parametrize("which", ["basis", "aux_basis"])
def test_formats_round_trip(case, writer, which):
    tab = golden(case)[which]
    assert inputs.parse_basis(writer(tab)) == tab


def test_sp_shell_is_split_s_then_p():
    tab = golden("631g2dfp")["basis"]
    assert any(len(a) == 2 for a, _, _ in blocks(tab["O"]))          # the fixture does hold sp shells
    got = inputs.parse_basis(to_gbs(tab))["O"]
    assert [s["l"] for s in got] == [s["l"] for s in tab["O"]]


def test_general_contraction_and_zero_coefficients():
    text = """basis "x" spherical
C  S
  10.0   0.5   0.0
   2.0   0.5   0.25
   0.5   0.0   1.0
C  P
   1.5   1.0
end
"""
    got = inputs.parse_basis(text)["C"]
    assert got == [{"l": 0, "exps": [10.0, 2.0], "coefs": [0.5, 0.5]},
                   {"l": 0, "exps": [2.0, 0.5], "coefs": [0.25, 1.0]},
                   {"l": 1, "exps": [1.5], "coefs": [1.0]}]


def test_gaussian94_scale_factor_and_fortran_exponents():
    text = "****\nH 0\nS 2 1.20\n 0.5D+01 0.3D+00\n 1.0D+00 0.7D+00\n****\n"
    got = inputs.parse_basis(text)["H"]
    assert got[0]["exps"] == pytest.approx([5.0 * 1.44, 1.44]) and got[0]["coefs"] == [0.3, 0.7]


def test_bad_tables_fail_loudly():
    with pytest.raises(ValueError):
        inputs.parse_basis("****\nH 0\nQ 1 1.0\n 1.0 1.0\n****\n")
    with pytest.raises(ValueError):
        inputs.parse_basis("****\nXx 0\nS 1 1.0\n 1.0 1.0\n****\n")
    with pytest.raises(ValueError):
        inputs.parse_bse_json({"elements": {"1": {"electron_shells": [
            {"angular_momentum": [0, 1], "exponents": ["1.0"], "coefficients": [["1.0"]]}]}}})


def write_case(tmp_path, case="ccpvdz", names=("cc-pVDZ", "cc-pVDZ-RIFIT"), scf=None):
    g = golden(case)
    geom = [x * inputs.ANGSTROM_PER_BOHR for a in g["atoms"] for x in a["center"]]
    doc = {"molecule": {"geometry": geom, "symbols": [a["symbol"] for a in g["atoms"]], "molecular_charge": 0,
                        "ignored": 1},
           "driver": "energy", "model": {"method": "RHF", "basis": names[0], "auxiliary_basis": names[1]},
           "keywords": {"scf": scf or {"niter": 50, "dele": 1e-6, "rmsd": 1e-6, "scf_type": "df",
                                       "contraction_mode": "GPU"}}}
    p = tmp_path / "water.json"
    p.write_text(json.dumps(doc))
    (tmp_path / (names[0].lower() + ".gbs")).write_text(to_gbs(g["basis"]))
    (tmp_path / (names[1] + ".nw")).write_text(to_nw(g["aux_basis"]))
    return str(p), g


def test_read_input_and_units(tmp_path):
    p, g = write_case(tmp_path)
    molecule, driver, model, keywords = inputs.read_input(p)
    assert sorted(molecule) == ["geometry", "molecular_charge", "symbols"]       # JCInput.jl:66-68 copies these three
    assert driver == "energy" and model["basis"] == "cc-pVDZ" and keywords["scf"]["niter"] == 50
    atoms = inputs.molecule_atoms(molecule)
    ref = np.array([a["center"] for a in g["atoms"]])
    assert np.allclose(np.array([a["center"] for a in atoms]), ref, rtol=0, atol=1e-14)
    assert [a["symbol"] for a in atoms] == [a["symbol"] for a in g["atoms"]]


def test_library_lookup(tmp_path, monkeypatch):
    p, g = write_case(tmp_path)
    molecule, _, model, _ = inputs.read_input(p)
    prim, aux = inputs.basis_tables(molecule, model, inputs.BasisLibrary([str(tmp_path)]))
    assert prim == g["basis"] and aux == g["aux_basis"]
    monkeypatch.setenv("JCDF_BASIS_PATH", str(tmp_path))
    assert inputs.BasisLibrary().get("cc-pVDZ") == g["basis"]
    with pytest.raises(FileNotFoundError):
        inputs.BasisLibrary([str(tmp_path)]).get("def2-SVP")
    del model["auxiliary_basis"]
    with pytest.raises(ValueError):
        inputs.basis_tables(molecule, model, inputs.BasisLibrary([str(tmp_path)]))
    with pytest.raises(KeyError):
        inputs.basis_tables({"symbols": ["C"]}, {"basis": "cc-pVDZ", "auxiliary_basis": "cc-pVDZ-RIFIT"},
                            inputs.BasisLibrary([str(tmp_path)]))


def test_xyz(tmp_path):
    p = tmp_path / "m.xyz"
    p.write_text("3\nwater\nO 0.0 -0.0757 0.0\nH 0.8668 0.6014 0.0\nH -0.8668 0.6014 0.0\n\n")
    coords, symbols = inputs.xyz_to_geometry(str(p))
    assert symbols == ["O", "H", "H"] and coords[3:6] == [0.8668, 0.6014, 0.0]
    assert inputs.xyz_to_molecule(str(p), 1)["molecular_charge"] == 1
    p.write_text("4\nwater\nO 0.0 -0.0757 0.0\n")
    with pytest.raises(ValueError):
        inputs.xyz_to_geometry(str(p))


def test_unsupported_requests_fail_loudly(tmp_path):
    p, _ = write_case(tmp_path)
    d = json.load(open(p))
    d["driver"] = "gradient"
    open(p, "w").write(json.dumps(d))
    with pytest.raises(ValueError):
        inputs.run_input(p, inputs.BasisLibrary([str(tmp_path)]))
    d["driver"], d["model"]["method"] = "energy", "MP2"
    open(p, "w").write(json.dumps(d))
    with pytest.raises(ValueError):
        inputs.run_input(p, inputs.BasisLibrary([str(tmp_path)]))


@pytest.mark.gpu
@pytest.mark.parametrize("case,names,niter,tol", [("ccpvdz", ("cc-pVDZ", "cc-pVDZ-RIFIT"), 50, 1e-9),
                                                  ("631g2dfp", ("6-31G(2df,p)", "cc-pVTZ-JKFIT"), 20, 1e-7)])
def test_input_file_end_to_end(tmp_path, case, names, niter, tol):
    """Input file in the reference's format + tables in the published text formats -> the energy of the reference's log."""
    p, g = write_case(tmp_path, case, names)           # the reference's thresholds for these logs: dele = rmsd = 1e-6
    out = inputs.run_input(p, inputs.BasisLibrary([str(tmp_path)]), scf_overrides={"niter": niter})
    assert out["Converged?"]
    assert abs(out["Energy"] - g["final_energy"]) < tol, out["Energy"]
