"""CPU oracle of the AvatarCap hot path -- test infrastructure only (see avatarcap_oracle.py)."""
