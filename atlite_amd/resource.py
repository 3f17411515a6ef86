"""
Turbine / panel configuration lookup - host-side inputs of the wind and pv kernels.

Mirrors the parts of atlite/resource.py the hot path calls: ``get_windturbineconfig``
(resource.py:50-109) incl. validation / cut-out padding (304-372), ``windturbine_smooth``
(227-297) and ``get_solarpanelconfig`` (112-141).  Configs come from the consolidated table
``resources/technologies.yaml`` or from a yaml file in the reference's per-turbine layout
(keys ``V``, ``POW``, ``HUB_HEIGHT``) when a ``pathlib.Path`` is given.  The OEDB download
(``"oedb:<name>"``, resource.py:375-509) needs network access and is out of scope.
"""

from __future__ import annotations

import logging
from pathlib import Path

import numpy as np
import yaml

logger = logging.getLogger(__name__)

_TABLE = Path(__file__).parent / "resources" / "technologies.yaml"
_Loader = getattr(yaml, "CSafeLoader", yaml.SafeLoader)
_text = None
_sections = {}


class _Section:
    """One top-level section of the consolidated table (``windturbine``, ``solarpanel``); an entry is parsed when it is
    asked for: the first ``cutout.pv()`` of a process needs one panel, the first ``cutout.wind()`` one power curve - parsing
    all thirty was 8 of a cold call's 17 ms (14 ms for the turbines)."""

    def __init__(self, kind, text):
        import re

        self._kind, self._text, self._rows = kind, text, {}
        marks = [(m.start(), m.group(1)) for m in re.finditer(r"^  ([^\s:#][^:\n]*):", text, re.M)]
        self._span = {name: (a, marks[i + 1][0] if i + 1 < len(marks) else len(text)) for i, (a, name) in enumerate(marks)}

    def __getitem__(self, name):
        if name not in self._rows:
            a, b = self._span[name]  # KeyError: no such entry
            self._rows[name] = yaml.load(self._text[a:b], Loader=_Loader)[name]  # libyaml's parser when present
        return self._rows[name]

    def __contains__(self, name):
        return name in self._span

    def __iter__(self):
        return iter(self._span)

    def __len__(self):
        return len(self._span)

    def keys(self):
        return self._span.keys()


class _Table:
    def __getitem__(self, kind):
        global _text
        if kind not in _sections:
            import re

            if _text is None:
                _text = _TABLE.read_text()
            starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(\w+):[ \t]*$", _text, re.M)]
            for i, (a, name) in enumerate(starts):
                if name == kind:
                    b = starts[i + 1][0] if i + 1 < len(starts) else len(_text)
                    _sections[kind] = _Section(kind, _text[_text.index("\n", a) + 1:b])
                    break
            else:
                raise KeyError(kind)
        return _sections[kind]


def _table():
    return _Table()


class _Catalogue(dict):
    """The reference's ``atlite.windturbines`` / ``atlite.solarpanels`` (resource.py:514-515: an attribute-access dict
    name -> yaml path, ``atlite.windturbines.Vestas_V112_3MW``).  Here the values are the names themselves - what
    ``get_windturbineconfig`` / ``get_solarpanelconfig`` and the ``turbine=`` / ``panel=`` arguments take - filled on first
    use from the one consolidated table; calling it lists the names."""

    def __init__(self, kind):
        super().__init__()
        self._kind, self._full = kind, False

    def _fill(self):
        if not self._full:
            self._full = True
            for k in _table()[self._kind]:
                dict.__setitem__(self, k, k)
        return self

    def __getattr__(self, item):
        if item.startswith("_"):
            raise AttributeError(item)
        try:
            return self._fill()[item]
        except KeyError as e:
            raise AttributeError(e.args[0])

    def __missing__(self, key):
        if not self._full and key in self._fill():
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __call__(self):
        return sorted(self._fill())

    def __dir__(self):
        return [k for k in self._fill() if k.isidentifier()]

    def __iter__(self):
        return dict.__iter__(self._fill())

    def __len__(self):
        return dict.__len__(self._fill())

    def __contains__(self, key):
        return dict.__contains__(self._fill(), key)

    def keys(self):
        return dict.keys(self._fill())

    def values(self):
        return dict.values(self._fill())

    def items(self):
        return dict.items(self._fill())


windturbines = _Catalogue("windturbine")
solarpanels = _Catalogue("solarpanel")


def _strip_yaml(name):
    return name[: -len(".yaml")] if name.endswith(".yaml") else name


def get_windturbineconfig(turbine, add_cutout_windspeed=True):
    """dict(V, POW, hub_height, P) for a turbine name, yaml path or user dict."""
    if not isinstance(turbine, (str, Path, dict)):
        raise KeyError(f"`turbine` must be a str, pathlib.Path or dict, but is {type(turbine)}.")
    if isinstance(turbine, str) and turbine.startswith("oedb:"):
        raise NotImplementedError("OEDB turbine download needs network access (atlite/resource.py:375-509)")
    if isinstance(turbine, str):
        row = _table()["windturbine"][_strip_yaml(turbine)]
        conf = dict(V=np.array(row["V"]), POW=np.array(row["POW"]), hub_height=row["hub_height"],
                    P=np.max(row["POW"]))
    elif isinstance(turbine, Path):
        with open(turbine) as f:
            raw = yaml.load(f, Loader=_Loader)
        conf = dict(V=np.array(raw["V"]), POW=np.array(raw["POW"]), hub_height=raw["HUB_HEIGHT"],
                    P=np.max(raw["POW"]))
    else:
        conf = turbine
    return _validate_turbine_config_dict(conf, add_cutout_windspeed)


def get_solarpanelconfig(panel):
    """Panel constants for a panel name or a yaml path."""
    assert isinstance(panel, (str, Path))
    if isinstance(panel, str):
        return dict(_table()["solarpanel"][_strip_yaml(panel)])
    with open(panel) as f:
        return yaml.safe_load(f)


def solarpanel_rated_capacity_per_unit(panel):
    """Rated capacity of one unit of a capacity layout (atlite/resource.py:204-217): the panel efficiency per m^2 for the
    Huld model, one panel's capacity for bofinger."""
    if isinstance(panel, (str, Path)):
        panel = get_solarpanelconfig(panel)
    model = panel.get("model", "huld")
    if model == "huld":
        return panel["efficiency"]
    if model == "bofinger":
        return (panel["A"] + panel["B"] * 1000.0 + panel["C"] * np.log(1000.0)) * 1e3


def windturbine_rated_capacity_per_unit(turbine):
    """Rated power of one turbine in MW (atlite/resource.py:220-224)."""
    if isinstance(turbine, (str, Path)):
        turbine = get_windturbineconfig(turbine)
    return turbine["P"]


def _max_v_is_zero_pow(turbine):
    return np.any(turbine["POW"][turbine["V"] == turbine["V"].max()] == 0)


def _validate_turbine_config_dict(turbine, add_cutout_windspeed):
    """Format checks and optional cut-out padding of a turbine dict (resource.py:304-372)."""
    need = ("POW", "V", "P", "hub_height")
    if not all(k in turbine for k in need):
        raise ValueError(
            "turbine config dict needs at least the following keys: ['POW', 'V', 'P', "
            f"'hub_height']\nbut are currently: {list(turbine.keys())}"
        )
    if not all(isinstance(turbine[p], (np.ndarray, list)) for p in ("POW", "V")):
        raise ValueError("turbine entries 'POW' and 'V' must be np.ndarray or list")
    if any(isinstance(turbine[p], list) for p in ("POW", "V")):
        turbine["V"] = np.array(turbine["V"])
        turbine["POW"] = np.array(turbine["POW"])
    if len(turbine["POW"]) != len(turbine["V"]):
        raise ValueError("turbine wind speed and power arrays do not have equal length.")
    if not np.all(np.diff(turbine["V"]) >= 0):
        # `>=`: cut-in / cut-out steps are stored as two knots at the same speed
        raise ValueError(
            "wind speed 'V' in the turbine config dict is expected to be increasing, "
            f"but is currently not in ascending order:\n{turbine['V']}"
        )
    if add_cutout_windspeed is True and not _max_v_is_zero_pow(turbine):
        turbine["V"] = np.pad(turbine["V"], (0, 1), "maximum")
        turbine["POW"] = np.pad(turbine["POW"], (0, 1), "constant", constant_values=0)
        logger.info(f"adding a cut-out wind speed to the turbine power curve at V={turbine['V'][-1]} m/s.")
    if not _max_v_is_zero_pow(turbine):
        logger.warning(
            "The power curve does not have a cut-out wind speed, i.e. the power"
            " output corresponding to the\nhighest wind speed is not zero. You can"
            " either change the power curve manually or set\n"
            "'add_cutout_windspeed=True' in the Cutout.wind conversion method."
        )
    return turbine


def windturbine_smooth(turbine, params=None):
    """
    Gaussian smoothing of the power curve (Andresen et al. 2015), resource.py:227-297: the
    curve is resampled on linspace(-50, 50, 1001), convolved with N(Delta_v, sigma) and
    resampled on linspace(0, 35, 72), scaled by the fleet availability eta.
    """
    from scipy.signal import fftconvolve

    if params is None or params is True:
        params = {}
    eta = params.get("eta", 0.95)
    Delta_v = params.get("Delta_v", 1.27)
    sigma = params.get("sigma", 2.29)

    v_reg = np.linspace(-50.0, 50.0, 1001)
    p_reg = np.interp(v_reg, turbine["V"], turbine["POW"])
    k_reg = 1.0 / np.sqrt(2 * np.pi * sigma * sigma) * np.exp(-(v_reg - Delta_v) * (v_reg - Delta_v) / (2 * sigma * sigma))
    conv = 0.1 * fftconvolve(p_reg, k_reg, mode="same")  # 0.1 = grid spacing
    v_new = np.linspace(0.0, 35.0, 72)
    p_new = eta * np.interp(v_new, v_reg, conv)

    out = turbine.copy()
    out["V"], out["POW"] = v_new, p_new
    out["P"] = np.max(p_new)
    if any(out["POW"][np.where(out["V"] == 0.0)] > 1e-2):
        logger.warning(
            "Oversmoothing detected with parameters eta=%f, Delta_v=%f, sigma=%f. "
            "Turbine generates energy at 0 m/s wind speeds.", eta, Delta_v, sigma)
    return out
