"""Test infrastructure only: CPU oracle for the t-SVGP E-step.  See tsvgp_oracle.py."""
