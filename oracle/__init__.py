"""CPU oracle for the SDNQ hot path -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.py)."""
