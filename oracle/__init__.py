"""CPU oracle (test infrastructure only — see oracle/oracle.cpp header)."""
