"""Input side of the DF-RHF path: the reference's JSON input files and basis-set tables, read into what rhf.run takes.

Mirrors (behaviour, not code):
  - JCInput.run                src/io/JCInput.jl:34-82        input file -> (molecule, driver, model, keywords)
  - xyz_to_geometry            src/io/xyz_to_molecule.jl:3-29 .xyz file -> flat coordinate list + symbols
  - JCBasis.run                src/basis/JCBasis.jl:42-61     geometry in Angstrom -> bohr with 1/0.52917724924,
                                                              shells per atom in input order, sp ("L") shells
                                                              split into their s and p parts
The reference looks basis sets up by name in an HDF5 table generated from the Basis Set Exchange at package-build
time (bsed.h5: needs network, and HDF5 is not in this image).  Here the tables are read from the text formats
the Basis Set Exchange itself serves — its JSON schema, Gaussian94 (.gbs) and NWChem (.nw) — found by name in the
directories given (or in $JCDF_BASIS_PATH).  Coefficients keep the convention of those files (they refer to normalised
primitives), the one include/jcint.h consumes.  Cartesian functions throughout, as in the reference."""
import json
import os
import re
from typing import Any, Dict, List, Optional, Sequence, Tuple

ANGSTROM_PER_BOHR = 0.52917724924           # JCBasis.jl:61 (the reference's constant, kept digit for digit)

_SYMBOLS = ("H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr "
            "Rb Sr Y Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs").split()
ATOMIC_NUMBER = {s: i + 1 for i, s in enumerate(_SYMBOLS)}      # 55 elements, like JCBasis.jl:77 (shells_cxx 1:55)
_AM = {"S": 0, "P": 1, "D": 2, "F": 3, "G": 4, "H": 5, "I": 6}


def canonical_symbol(s: str) -> str:
    s = s.strip()
    s = s[:1].upper() + s[1:].lower()
    if s not in ATOMIC_NUMBER:
        raise ValueError("unknown element symbol %r" % s)
    return s


# ------------------------------------------------------------------------------------------------ input files
def read_input(path: str) -> Tuple[Dict[str, Any], str, Dict[str, Any], Dict[str, Any]]:
    """JCInput.run (JCInput.jl:34-82): (molecule, driver, model, keywords); molecule keeps exactly the three keys the
    reference copies (geometry: flat list in Angstrom, symbols, molecular_charge)."""
    with open(path) as f:
        d = json.load(f)
    mol = d["molecule"]
    molecule = {"geometry": [float(x) for x in mol["geometry"]], "symbols": list(mol["symbols"]),
                "molecular_charge": mol["molecular_charge"]}
    if len(molecule["geometry"]) != 3 * len(molecule["symbols"]):
        raise ValueError("%s: %d coordinates for %d symbols" % (path, len(molecule["geometry"]), len(molecule["symbols"])))
    return molecule, d["driver"], dict(d["model"]), dict(d["keywords"])


def xyz_to_geometry(path: str) -> Tuple[List[float], List[str]]:
    """xyz_to_molecule.jl:3-29: line 1 = atom count, line 2 = comment, then `symbol x y z` (Angstrom)."""
    with open(path) as f:
        lines = f.read().splitlines()
    natoms = int(lines[0].split()[0])
    coords: List[float] = []
    symbols: List[str] = []
    for line in lines[2:]:
        t = line.split()
        if not t:
            continue
        symbols.append(t[0])
        coords += [float(x) for x in t[1:4]]
    if len(symbols) != natoms:
        raise ValueError("%s: header says %d atoms, %d found" % (path, natoms, len(symbols)))
    return coords, symbols


def xyz_to_molecule(path: str, charge: int = 0) -> Dict[str, Any]:
    coords, symbols = xyz_to_geometry(path)
    return {"geometry": coords, "symbols": symbols, "molecular_charge": charge}


def molecule_atoms(molecule: Dict[str, Any]) -> List[Dict[str, Any]]:
    """The atom list rhf.run takes: centres in bohr (JCBasis.jl:57-61)."""
    g = molecule["geometry"]
    f = 1.0 / ANGSTROM_PER_BOHR
    return [{"symbol": canonical_symbol(s), "center": [g[3 * i] * f, g[3 * i + 1] * f, g[3 * i + 2] * f]}
            for i, s in enumerate(molecule["symbols"])]


# ------------------------------------------------------------------------------------------------ basis tables
def _num(tok: str) -> float:
    return float(tok.replace("D", "E").replace("d", "e"))


def _emit(shells: List[Dict], ams: Sequence[int], exps: Sequence[float], cols: Sequence[Sequence[float]]) -> None:
    """One contracted block -> shells.  len(ams) > 1: one coefficient column per angular momentum (sp shells: s first,
    then p, the order the reference adds them in).  One angular momentum and several columns: a general contraction,
    one shell per column.  Primitives whose coefficient is exactly zero are dropped."""
    if len(ams) > 1:
        if len(cols) != len(ams):
            raise ValueError("combined shell with %d momenta and %d coefficient columns" % (len(ams), len(cols)))
        pairs = list(zip(ams, cols))
    else:
        pairs = [(ams[0], c) for c in cols]
    for l, col in pairs:
        if len(col) != len(exps):
            raise ValueError("coefficient column of length %d for %d exponents" % (len(col), len(exps)))
        keep = [k for k, c in enumerate(col) if c != 0.0]
        if keep:
            shells.append({"l": int(l), "exps": [float(exps[k]) for k in keep], "coefs": [float(col[k]) for k in keep]})


def _ams(code: str) -> List[int]:
    code = code.upper()
    if code == "L":
        code = "SP"
    try:
        return [_AM[c] for c in code]
    except KeyError:
        raise ValueError("unknown shell type %r" % code)


def parse_bse_json(d: Dict[str, Any]) -> Dict[str, List[Dict]]:
    """Basis Set Exchange JSON (molssi_bse_schema): elements -> electron_shells."""
    out: Dict[str, List[Dict]] = {}
    for z, el in d["elements"].items():
        shells: List[Dict] = []
        for sh in el.get("electron_shells", []):
            if not sh.get("function_type", "gto").startswith("gto"):
                raise ValueError("only Gaussian shells (function_type %r)" % sh.get("function_type"))
            exps = [_num(x) for x in sh["exponents"]]
            cols = [[_num(x) for x in col] for col in sh["coefficients"]]
            _emit(shells, [int(a) for a in sh["angular_momentum"]], exps, cols)
        out[_SYMBOLS[int(z) - 1]] = shells
    return out


def parse_gaussian94(text: str) -> Dict[str, List[Dict]]:
    """Gaussian94 (.gbs): `****`-separated elements, `Sym 0`, then blocks `TYPE nprim scale` + nprim rows."""
    out: Dict[str, List[Dict]] = {}
    lines = [ln.split("!")[0].strip() for ln in text.splitlines()]
    lines = [ln for ln in lines if ln]
    i = 0
    sym: Optional[str] = None
    while i < len(lines):
        t = lines[i].split()
        if lines[i].startswith("****"):
            sym = None
            i += 1
        elif sym is None:
            sym = canonical_symbol(t[0].lstrip("-"))
            out.setdefault(sym, [])
            i += 1
        else:
            ams, nprim = _ams(t[0]), int(t[1])
            scale = _num(t[2]) if len(t) > 2 else 1.0
            rows = [[_num(x) for x in lines[i + 1 + k].split()] for k in range(nprim)]
            exps = [r[0] * scale * scale for r in rows]
            cols = [[r[1 + c] for r in rows] for c in range(len(rows[0]) - 1)]
            _emit(out[sym], ams, exps, cols)
            i += 1 + nprim
    return out


def parse_nwchem(text: str) -> Dict[str, List[Dict]]:
    """NWChem (.nw): `basis ... end` with blocks `Sym TYPE` + rows `exponent c1 [c2 ...]`."""
    out: Dict[str, List[Dict]] = {}
    cur: Optional[Tuple[str, List[int]]] = None
    rows: List[List[float]] = []

    def flush() -> None:
        if cur is not None and rows:
            cols = [[r[1 + c] for r in rows] for c in range(len(rows[0]) - 1)]
            _emit(out.setdefault(cur[0], []), cur[1], [r[0] for r in rows], cols)

    inside = False
    for raw in text.splitlines():
        ln = raw.split("#")[0].strip()
        if not ln:
            continue
        low = ln.lower()
        if low.startswith("basis"):
            inside = True
            continue
        if low == "end":
            flush()
            cur, rows, inside = None, [], False
            continue
        if not inside:
            continue
        t = ln.split()
        if re.match(r"^[A-Za-z]{1,2}$", t[0]) and len(t) == 2 and re.match(r"^[A-Za-z]+$", t[1]):
            flush()
            cur, rows = (canonical_symbol(t[0]), _ams(t[1])), []
        else:
            rows.append([_num(x) for x in t])
    flush()
    return out


def parse_basis(text: str) -> Dict[str, List[Dict]]:
    """Any of the three formats, told apart by content."""
    s = text.lstrip()
    if s.startswith("{"):
        return parse_bse_json(json.loads(text))
    if re.search(r"^\s*basis\b", text, re.I | re.M) and re.search(r"^\s*end\s*$", text, re.I | re.M):
        return parse_nwchem(text)
    return parse_gaussian94(text)


def load_basis(path: str) -> Dict[str, List[Dict]]:
    with open(path) as f:
        return parse_basis(f.read())


class BasisLibrary:
    """Basis tables by name, as the reference's bsed.h5 lookup (JCBasis.jl:98-150) — here from files
    `<name>.json|.gbs|.nw` (exact name, lower case, or the Basis Set Exchange's file-name form) in the given directories,
    then in $JCDF_BASIS_PATH."""

    def __init__(self, dirs: Sequence[str] = ()):
        env = [p for p in os.environ.get("JCDF_BASIS_PATH", "").split(os.pathsep) if p]
        self.dirs = [str(d) for d in dirs] + env
        self._cache: Dict[str, Dict[str, List[Dict]]] = {}

    @staticmethod
    def _file_names(name: str) -> List[str]:
        bse = name.lower().replace("*", "_st_").replace("/", "_sl_").replace("(", "_").replace(")", "_").replace(" ", "_")
        stems = []
        for s in (name, name.lower(), bse):
            if s not in stems:
                stems.append(s)
        return [s + ext for s in stems for ext in (".json", ".gbs", ".nw")]

    def get(self, name: str) -> Dict[str, List[Dict]]:
        if name in self._cache:
            return self._cache[name]
        for d in self.dirs:
            for fn in self._file_names(name):
                p = os.path.join(d, fn)
                if os.path.isfile(p):
                    self._cache[name] = load_basis(p)
                    return self._cache[name]
        raise FileNotFoundError("basis set %r not found in %s (set JCDF_BASIS_PATH, or pass the table itself)"
                                % (name, self.dirs or "[no directories given]"))


def basis_tables(molecule: Dict[str, Any], model: Dict[str, Any], library: Optional[BasisLibrary] = None,
                 basis: Optional[Dict[str, List[Dict]]] = None, aux_basis: Optional[Dict[str, List[Dict]]] = None):
    """JCBasis.run's lookup part (JCBasis.jl:52-56): model["basis"] and model["auxiliary_basis"], restricted to the
    elements of the molecule.  Explicit tables win over the library."""
    lib = library or BasisLibrary()
    if basis is None:
        basis = lib.get(model["basis"])
    if aux_basis is None:
        if "auxiliary_basis" not in model:
            raise ValueError("density fitting needs model[\"auxiliary_basis\"] (JCBasis.jl:55)")
        aux_basis = lib.get(model["auxiliary_basis"])
    els = sorted({canonical_symbol(s) for s in molecule["symbols"]}, key=ATOMIC_NUMBER.get)
    for name, tab in (("basis", basis), ("auxiliary basis", aux_basis)):
        missing = [e for e in els if not tab.get(e)]
        if missing:
            raise KeyError("%s has no shells for %s" % (name, ", ".join(missing)))
    return {e: basis[e] for e in els}, {e: aux_basis[e] for e in els}


def run_input(path: str, library: Optional[BasisLibrary] = None, basis: Optional[Dict[str, List[Dict]]] = None,
              aux_basis: Optional[Dict[str, List[Dict]]] = None, output: int = 0, device: Optional[int] = None,
              scf_overrides: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    """An input file through the whole path, the sequence of the reference's driver scripts
    (example_scripts/minimal-rhf.jl: JCInput.run -> JCBasis.run -> JCRHF.Energy.run) with keywords["scf"] as the
    SCF flags.  Only driver "energy" / method "RHF" with scf_type "df" on the device exist here."""
    from . import rhf
    molecule, driver, model, keywords = read_input(path)
    if driver != "energy":
        raise ValueError("driver %r: only \"energy\" is implemented" % driver)
    if str(model.get("method", "RHF")).upper() != "RHF":
        raise ValueError("method %r: only RHF is implemented" % model.get("method"))
    flags = dict(keywords.get("scf", {}))
    flags.update(scf_overrides or {})
    prim, aux = basis_tables(molecule, model, library, basis, aux_basis)
    atoms = molecule_atoms(molecule)
    charges = {a["symbol"]: float(ATOMIC_NUMBER[a["symbol"]]) for a in atoms}
    return rhf.run(atoms, charges, prim, aux, scf_flags=flags, molecular_charge=int(molecule["molecular_charge"]),
                   output=output, device=device)
