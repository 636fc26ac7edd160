"""Residuals — Python mirror of Aggregation.jl:20-22."""
from __future__ import annotations


def Residuals(xMat, model):
    """evaluate the model's compiled equations on the padded n_v x T_pad matrix; returns the
    residual vector ordered all equations at t=1, then t=2, ... (Aggregation.jl:14-22)."""
    return model.residuals_fn(xMat, model.params)
