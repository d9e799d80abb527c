"""Support code of bench.py (repo root): launch plumbing, measurement legs, the CPU baseline."""
