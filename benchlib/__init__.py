"""Helpers of bench.py (the driver with the timed region stays at the repo root): synthetic inputs and timing (common), roofline objects
(roofline), CPU baseline and parity report (baselines), the other BASELINE configs and pipelines (configs), the unchanged reference loop (loops)."""
