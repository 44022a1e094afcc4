"""CPU restatement (numpy, f32) of SplatTrainer::refine's deterministic part — test infrastructure only.

Follows brush-train/src/train.rs:
  prune mask            :485-524   (opacity < 1/255, scale > 100 x max extent, |mean - center| > same, non-finite)
  prune_points          :848-893   (stable gather of every tensor + RefineRecord)
  refine_splats         :665-822   (covariance-aware split, moment reset, opacity decay)
  bounds_from_pos       brush-train/src/splat_init.rs:130-160
The stochastic choices (which indices multinomial_sample returns, train.rs:539,621; the reference uses
an unseeded rand::rng()) are INPUTS here: the parity test feeds the indices the HIP plan chose and checks
everything downstream, plus the counting rules that bound how many may be chosen.
"""
import math

import numpy as np

F = np.float32
MIN_OPACITY = F(1.0 / 255.0)
FRAC_1_SQRT_2 = F(0.70710678118654752440)


def sigmoid(x):
    x = np.asarray(x, F)
    return (F(1.0) / (F(1.0) + np.exp(-x, dtype=F))).astype(F)


def inv_sigmoid(x):
    x = np.asarray(x, F)
    return np.log(x / (F(1.0) - x), dtype=F).astype(F)


def prune_mask(transforms, sh, raw_opac, bounds_center, bounds_extent):
    with np.errstate(all="ignore"):
        max_allowed = F(max(bounds_extent)) * F(100.0)
        alpha = sigmoid(raw_opac) < MIN_OPACITY
        scale_big = (np.exp(transforms[:, 7:10], dtype=F) > max_allowed).any(axis=1)
        bound = (np.abs(transforms[:, 0:3] - np.asarray(bounds_center, F)[None, :]) > max_allowed).any(axis=1)
        bad = (~np.isfinite(transforms)).any(axis=1) | (~np.isfinite(sh.reshape(sh.shape[0], -1))).any(axis=1) | ~np.isfinite(raw_opac)
    return alpha | scale_big | bound | bad, bad


def budget_counts(n, keep, vis_weight, refine_norm, max_screen, cfg):
    """How many indices each stage may select (train.rs:527-623); returns dict of the scalars."""
    n_keep = int(keep.sum())
    pruned = n - n_keep
    vis = vis_weight > 0
    oversized = keep & vis & (max_screen > F(cfg["split_at_screen_size"])) if cfg["split_at_screen_size"] > 0 else np.zeros(n, bool)
    above = keep & vis & (refine_norm > F(cfg["growth_grad_threshold"]))
    thr = int(above.sum())
    want = F(thr) * F(cfg["growth_select_fraction"])
    grow = int(math.floor(abs(float(want)) + 0.5)) if want > 0 else 0  # f32::round, half away from zero
    return dict(n_keep=n_keep, pruned=pruned, oversized=oversized, above=above, threshold_count=thr, grow_count=grow)


def apply(state, keep, split, cfg, max_screen):
    """state: dict of numpy arrays (transforms, sh, raw_opac, m1_t, m2_t, m1_sh, m2_sh, m1_o, m2_o);
    keep/split: bool [n] (split subset of keep).  Children are appended in ascending parent order."""
    with np.errstate(all="ignore"):
        idx = np.nonzero(keep)[0]
        out = {k: v[idx].copy() for k, v in state.items()}
        new_row = np.cumsum(keep) - 1
        parents_old = np.nonzero(split)[0]
        par = new_row[parents_old]
        r = len(par)
        if r:
            t = out["transforms"][par]
            cur_means, rots_raw, cur_log = t[:, 0:3], t[:, 3:7], t[:, 7:10]
            mag = np.maximum(np.sqrt((rots_raw * rots_raw).sum(axis=1, dtype=F), dtype=F), F(1e-32))
            rots = (rots_raw / mag[:, None]).astype(F)
            scales = np.exp(cur_log, dtype=F)
            raw = out["raw_opac"][par]
            inv_opac = F(1.0) - sigmoid(raw)
            new_opac = F(1.0) - np.power(inv_opac, FRAC_1_SQRT_2, dtype=F)
            new_raw = inv_sigmoid(np.clip(new_opac, MIN_OPACITY, F(1.0) - MIN_OPACITY))
            sq = scales * scales
            max_sq = np.maximum(sq.max(axis=1), F(1e-30))
            ratio = sq / max_sq[:, None]
            if cfg["split_at_screen_size"] > 0:
                k_max = np.minimum((F(1.0) / np.maximum(max_screen[parents_old], F(1e-6))) * F(cfg["split_at_screen_size"]), FRAC_1_SQRT_2)[:, None]
            else:
                k_max = FRAC_1_SQRT_2
            k_axis = (-(ratio * (-k_max + F(1.0))) + F(1.0)).astype(F)
            off = (np.sqrt(np.maximum(-(k_axis * k_axis) + F(1.0), F(0.0)), dtype=F) * scales).astype(F)
            qw, qx, qy, qz = rots[:, 0], rots[:, 1], rots[:, 2], rots[:, 3]
            vx, vy, vz = off[:, 0], off[:, 1], off[:, 2]
            qw2, qx2, qy2, qz2 = qw * qw, qx * qx, qy * qy, qz * qz
            xy, xz, yz, wx, wy, wz = qx * qy, qx * qz, qy * qz, qw * qx, qw * qy, qw * qz
            sx = (qw2 + qx2 - qy2 - qz2) * vx + (xy * vy + xz * vz + wy * vz - wz * vy) * F(2.0)
            sy = (qw2 - qx2 + qy2 - qz2) * vy + (xy * vx + yz * vz + wz * vx - wx * vz) * F(2.0)
            sz = (qw2 - qx2 - qy2 + qz2) * vz + (xz * vx + yz * vy + wx * vy - wy * vx) * F(2.0)
            samples = np.stack([sx, sy, sz], axis=1).astype(F)
            new_log = (cur_log + np.log(k_axis, dtype=F)).astype(F)
            child_t = np.concatenate([cur_means + samples, rots, new_log], axis=1).astype(F)
            out["transforms"][par, 0:3] = cur_means + (-samples)
            out["transforms"][par, 7:10] = cur_log + (new_log - cur_log)
            out["raw_opac"][par] = raw + (new_raw - raw)
            for k in ("m1_t", "m2_t", "m1_sh", "m2_sh", "m1_o", "m2_o"):
                out[k][par] = 0
            out["transforms"] = np.concatenate([out["transforms"], child_t])
            out["sh"] = np.concatenate([out["sh"], out["sh"][par]])
            out["raw_opac"] = np.concatenate([out["raw_opac"], new_raw])
            for k in ("m1_t", "m2_t", "m1_sh", "m2_sh", "m1_o", "m2_o"):
                out[k] = np.concatenate([out[k], np.zeros((r,) + out[k].shape[1:], F)])
        train_t = min(max(F(cfg["iter"]) / F(cfg["total_train_iters"]), F(0.0)), F(1.0))
        minus = F(cfg["opac_decay"]) * (F(1.0) - F(train_t))
        out["raw_opac"] = inv_sigmoid(np.clip(sigmoid(out["raw_opac"]) - minus, F(1e-12), F(1.0) - F(1e-12)))
    return out


def bounds_from_pos(percentile, means):
    """splat_init.rs:130-160 -> (center[3], extent[3])."""
    mn, mx = [], []
    for a in range(3):
        v = np.sort(means[:, a][np.isfinite(means[:, a])])
        if v.size == 0:
            return (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
        n = v.size
        pc = F(percentile)  # f32 arithmetic step by step, as the Rust expression evaluates
        lo = int((F(1.0) - pc) / F(2.0) * F(n))
        hi = min(n - 1, int((F(1.0) + pc) / F(2.0) * F(n)))
        mn.append(F(v[lo]))
        mx.append(F(v[hi]))
    mn, mx = np.array(mn, F), np.array(mx, F)
    return tuple((mx + mn) / F(2.0)), tuple((mx - mn) / F(2.0))
