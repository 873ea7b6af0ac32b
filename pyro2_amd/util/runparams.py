"""Runtime parameters: `[section]` / `key = value ; comment` files layered as
defaults -> problem parameters -> inputs file -> overrides.

Same public surface and file format as pyro/util/runparams.py:83-274
(RuntimeParameters.load_params / get_param / set_param / print_paramfile /
write_params ...), own implementation.
"""
import os
import re

from . import msg

_SECTION = re.compile(r"^\s*\[([^\]]*)\]")
_ASSIGN = re.compile(r"^([^=#;\[]+)=([^;]*)(?:;(.*))?$")


def is_int(string):
    try:
        int(string)
    except ValueError:
        return False
    return True


def is_float(string):
    try:
        float(string)
    except ValueError:
        return False
    return True


def _get_val(value):
    """int if it parses as int, else float, else the stripped string
    (pyro/util/runparams.py:76-81)"""
    for conv in (int, float):
        try:
            return conv(value)
        except ValueError:
            pass
    return value.strip()


class RuntimeParameters:
    def __init__(self):
        self.params = {}
        self.param_comments = {}
        self.used_params = []

    # -- reading ----------------------------------------------------------
    def load_params(self, pfile, *, no_new=False):
        """parse a parameter file.  With no_new only existing keys may be
        overridden; unknown ones are reported and skipped."""
        if not os.path.isfile(pfile):
            alt = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), pfile)
            if os.path.isfile(alt):
                pfile = alt
        try:
            with open(pfile) as f:
                lines = f.readlines()
        except OSError:
            msg.fail(f"ERROR: parameter file does not exist: {pfile}")
            return
        section = ""
        for raw in lines:
            line = raw.rstrip("\n")
            m = _SECTION.match(line)
            if m:
                section = m.group(1).strip().lower()
                continue
            m = _ASSIGN.match(line.strip())
            if not m:
                continue
            key = section + "." + m.group(1).strip().lower()
            comment = (m.group(3) or "").strip()
            if no_new and key not in self.params:
                msg.warning(f"warning, key: {key} not defined")
                continue
            self.params[key] = _get_val(m.group(2))
            if comment == "":
                comment = self.param_comments.get(key, "")
            self.param_comments[key] = comment

    # -- access -----------------------------------------------------------
    def _ensure_loaded(self):
        if not self.params:
            msg.warning("WARNING: runtime parameters not yet initialized")
            self.load_params("_defaults")

    def get_param(self, key):
        self._ensure_loaded()
        if key not in self.used_params:
            self.used_params.append(key)
        try:
            return self.params[key]
        except KeyError:
            raise KeyError(f"ERROR: runtime parameter {key} not found") from None

    def set_param(self, key, value, *, no_new=True):
        self._ensure_loaded()
        if key in self.params:
            self.params[key] = value
            if not no_new:
                self.param_comments[key] = ""
            return
        if no_new:
            raise KeyError(f"ERROR: runtime parameter {key} not found")
        self.params[key] = value
        self.param_comments[key] = ""

    # -- reporting --------------------------------------------------------
    def print_unused_params(self):
        for key in self.params:
            if key not in self.used_params:
                msg.warning(f"parameter {key} never used")

    def print_all_params(self):
        for key in sorted(self.params):
            print(key, "=", self.params[key])
        print(" ")

    def write_params(self, f):
        """store every parameter as an attribute of an HDF5 group
        "runtime parameters" (f is an h5py file object)"""
        grp = f.create_group("runtime parameters")
        for key in sorted(self.params):
            grp.attrs[key] = self.params[key]

    def __str__(self):
        return "".join(f"{k} = {self.params[k]}\n" for k in sorted(self.params))

    def print_paramfile(self, filename="inputs.auto"):
        """dump all parameters in inputs-file syntax"""
        sections = sorted({k.split(".", 1)[0] for k in self.params})
        try:
            f = open(filename, "w")
        except OSError:
            msg.fail(f"ERROR: unable to open {filename}")
            return
        with f:
            f.write("# automagically generated parameter file\n")
            for sec in sections:
                f.write(f"\n[{sec}]\n")
                for key, value in self.params.items():
                    if not key.startswith(sec + "."):
                        continue
                    opt = key.split(".", 1)[1]
                    com = self.param_comments.get(key, "")
                    f.write(f"{opt} = {value}    ; {com}\n" if com else f"{opt} = {value}\n")
