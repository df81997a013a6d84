"""TEST INFRASTRUCTURE ONLY (see oracle/rene_oracle.cpp). Importable from tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke(); never from rene_amd/."""
