"""Parity bookkeeping for the GPU tests: ``check(name, value, bound)`` asserts ``value < bound`` AND records the measured value,
so that every tolerance in the suite sits next to the number it was derived from (bounds are kept at <= 3x the measured value;
the record of a run is written to gpurun_out/parity_measured.json and a copy is committed under profiles/)."""
import atexit
import json
import os

_REC = {}
_OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_measured.json")


def check(name, value, bound, inclusive=False, note=""):
    """``note``: what the bound was derived from (e.g. the reference's own spread on the same case), quoted in the failure message."""
    value, bound = float(value), float(bound)
    _REC[name] = dict(measured=value, bound=bound, **({"note": note} if note else {}))
    ok = value <= bound if inclusive else value < bound
    assert ok, f"{name}: measured {value:.4g} exceeds the bound {bound:.4g}" + (f" [{note}]" if note else "")
    return value


@atexit.register
def _dump():
    if _REC:
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        with open(_OUT, "w") as f:
            json.dump(_REC, f, indent=1, sort_keys=True)
