"""CPU oracle of the bundle-adjustment hot path -- TEST INFRASTRUCTURE ONLY (see ba_oracle.py, lm_oracle.py)."""
