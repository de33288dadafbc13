"""CPU oracle (test infrastructure only — see angio_oracle.py header)."""
