"""Per-shape FLOOR of a W4A8 GEMM launch on MI355X: what the launch would take if its k-loop ran at the matrix rate and every
fixed cost were the hardware's own (VERDICT r5 "next" 2a).  bench.py puts the summed floor beside the measured ``frac``
(``roofline.frac_floor_model``) with the per-shape table, so that the distance between the two is a number, not an argument.

    floor(launch) = dispatch gap + wave launch + first-stage latency + rounds x k-steps x (MFMA cycles of a k-step / clock) + store tail

The constants are measurements of this repository on MI355X, each the SMALLEST value any shape showed (a floor must not be beatable):

* dispatch gap 1.23 us -- last workgroup of launch g done -> first workgroup of launch g+1, dependent launches from one hipGraph
  (profiles/r5_ws_fixed_cost_timeline.txt: 1.23-1.60 us per shape; /opt/skills/guides/MI355X_MICROARCH.md "boundary": 1.1-1.4 us inside a
  GEMM chain).  Reference op: every ActQuantWrapper.forward is its own Linear (fake_quant/quant_utils.py:330-391), so a launch per
  Linear is the path's own structure.
* wave launch 0.23 us -- first wave's entry -> the last (loader) wave's first instruction; the hardware starts the 8-12 waves of a
  workgroup one after the other (same file: 0.23-0.47 us).
* first stage 0.45 us -- one scalar-load round trip for the argument block (~0.12 us), the tile map, and the first operand stage
  from HBM / Infinity Cache (~900 cycles for an HBM miss, the guide's constant): nothing can be multiplied before it lands.
* k-step: BM x BN x 128 MACs on a CU whose four matrix pipes retire 4096 int8 MACs per clock (5 POP/s nominal = 256 CUs x 8192 ops x
  2.4 GHz), times the rounds ceil(tiles / CUs).
* store tail 0.40 us -- last MFMA -> the workgroup's stores acknowledged with NO epilogue arithmetic in between: accumulator read-out
  plus the store round trip (same file: issued -> acknowledged 0.14-0.18 us per wave, -> every wave 0.27-1.2 us).

The clock: a launch whose k-loop is matrix-bound runs at the package power limit (profiles/r5_clock_reconciliation.txt), i.e. at the
clock at which the box SUSTAINS int8 MFMAs on this workload's operand bytes -- bench.py measures that rate live
(``roofline.peak_sustained_measured``) and passes clock = 2.4 GHz x sustained / nominal.  ``frac_floor_model_nominal_clock`` repeats the
sum at 2.4 GHz (unreachable on real operand bytes; it bounds the model from the other side).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Tuple

#: tile id (csrc/gemm_w4a8.hip dispatch_tile / gemm_ws.hip dispatch_ws) -> (BM, BN)
TILE_SHAPES: Dict[int, Tuple[int, int]] = {
    1: (256, 256), 2: (256, 128), 3: (256, 256), 4: (128, 256), 5: (256, 128), 10: (64, 128), 11: (128, 64), 12: (128, 128),
    13: (256, 256), 14: (256, 256), 15: (128, 128), 16: (96, 128), 17: (192, 128), 18: (64, 128), 19: (128, 256), 20: (256, 256),
    26: (128, 128), 31: (96, 128), 35: (192, 128),
    40: (96, 128), 41: (128, 128), 42: (192, 128), 43: (64, 128), 44: (96, 128), 45: (128, 128), 46: (192, 128), 47: (64, 128), 48: (96, 128),
    50: (96, 128), 51: (128, 128), 52: (192, 128), 53: (64, 128), 54: (96, 128),
}

GAP_US = 1.23
WAVE_LAUNCH_US = 0.23
FIRST_STAGE_US = 0.45
STORE_TAIL_US = 0.40
MAC_PER_CLK_CU = 4096.0
NOMINAL_GHZ = 2.4
NOMINAL_TOPS = 5000.0


def fixed_us() -> float:
    return GAP_US + WAVE_LAUNCH_US + FIRST_STAGE_US + STORE_TAIL_US


def launch_floor_us(M: int, N: int, K_pad: int, tile: int, clock_ghz: float, cus: int = 256) -> float:
    """Floor of ONE launch of an M x N x K_pad GEMM on tile ``tile`` at ``clock_ghz`` (see the module docstring)."""
    BM, BN = TILE_SHAPES.get(tile, (128, 128))
    tiles = math.ceil(M / BM) * math.ceil(N / BN)
    rounds = math.ceil(tiles / cus)
    kstep_cycles = BM * BN * 128 / MAC_PER_CLK_CU
    return fixed_us() + rounds * (K_pad / 128.0) * kstep_cycles / (clock_ghz * 1e3)


def plan_tile(M: int, N: int, K_pad: int, w_bits: int = 4) -> int:
    """The tile the dispatcher takes for this shape (``mq_gemm_debug_plan``: host arithmetic, no device needed)."""
    import ctypes as C

    from . import _lib
    tile, splits = C.c_int(0), C.c_int(0)
    rc = _lib.load().mq_gemm_debug_plan(M, N, K_pad, w_bits, 1, 1, C.byref(tile), C.byref(splits))
    if rc != 0:
        raise _lib.MQuantHipError("mq_gemm_debug_plan failed")
    return int(tile.value)


def shape_groups(layers: Iterable) -> List[dict]:
    """The distinct GEMM shapes of a prefill object (``workload._HotPath.layers``) in model order: name, M, N, K_pad, w_bits, the
    layers of that shape."""
    groups: Dict[tuple, dict] = {}
    for L in layers:
        key = (L.spec.M, L.lin.N, L.lin.K_pad, L.lin.w_bits)
        g = groups.get(key)
        if g is None:
            g = groups[key] = {"name": getattr(L, "order_name", L.spec.name), "M": L.spec.M, "N": L.lin.N, "K_pad": L.lin.K_pad,
                               "w_bits": L.lin.w_bits, "layers": []}
        g["layers"].append(L)
    return list(groups.values())


def summarize(groups: List[dict], measured_us: Dict[int, float], sustained_tops: float, cus: int = 256) -> dict:
    """Per-shape table + the summed floor.  ``measured_us[i]`` = average launch time of group i, measured by the caller from a
    hipGraph of that group's launches back to back (the dispatch gap is inside it, as it is inside the step)."""
    clock = NOMINAL_GHZ * min(1.0, sustained_tops / NOMINAL_TOPS) if sustained_tops else NOMINAL_GHZ
    rows, floor_total, floor_nom, meas_total, ops_total = [], 0.0, 0.0, 0.0, 0.0
    for i, g in enumerate(groups):
        n = len(g["layers"])
        tile = plan_tile(g["M"], g["N"], g["K_pad"], g["w_bits"])
        fl = launch_floor_us(g["M"], g["N"], g["K_pad"], tile, clock, cus)
        fl_nom = launch_floor_us(g["M"], g["N"], g["K_pad"], tile, NOMINAL_GHZ, cus)
        ops = 2.0 * g["M"] * g["N"] * g["K_pad"]
        us = measured_us.get(i)
        rows.append({"shape": f"{g['name']} {g['M']}x{g['N']}x{g['K_pad']}" + (" W8" if g["w_bits"] == 8 else ""), "launches": n, "tile": tile,
                     "us": None if us is None else round(us, 2), "floor_us": round(fl, 2),
                     "frac": None if us is None else round(ops / (us * 1e-6) / 1e12 / NOMINAL_TOPS, 4),
                     "frac_at_floor": round(ops / (fl * 1e-6) / 1e12 / NOMINAL_TOPS, 4)})
        floor_total += n * fl
        floor_nom += n * fl_nom
        ops_total += n * ops
        if us is not None:
            meas_total += n * us
    return {"per_shape": rows, "floor_ms_per_step": round(floor_total * 1e-3, 4),
            "frac_floor_model": round(ops_total / (floor_total * 1e-6) / 1e12 / NOMINAL_TOPS, 4),
            "frac_floor_model_nominal_clock": round(ops_total / (floor_nom * 1e-6) / 1e12 / NOMINAL_TOPS, 4),
            "per_shape_measured_ms_per_step": round(meas_total * 1e-3, 4),
            "clock_ghz": round(clock, 3), "fixed_us_per_launch": round(fixed_us(), 2),
            "model": "floor = 1.23 us dispatch gap + 0.23 wave launch + 0.45 first stage + rounds x k-steps x (BM x BN x 128 MACs / 4096 per clock and CU) "
                     "/ clock + 0.40 store tail; clock = 2.4 GHz x sustained / nominal int8 rate of this box (mquant_amd/floor_model.py)"}
