"""CPU oracle for the extract hot path - TEST INFRASTRUCTURE ONLY (see oracle/README.md)."""
