"""Tensor-parallel process-group state and collectives (one process per GPU, RCCL over xGMI)."""
