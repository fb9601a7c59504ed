def lp_loss(pred, tgt, p=2.0, reduction="none"):
    """L_p distance; ``reduction="none"`` sums over dim 1 then averages, anything else is a
    plain mean (the naming is upstream's)."""
    err = (pred - tgt).abs().pow(p)
    return err.sum(1).mean() if reduction == "none" else err.mean()
