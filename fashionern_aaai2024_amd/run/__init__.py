"""Recall@K harness: counterparts of /root/reference/run/test/test_{fiq,cirr,200k,shoes,val}.py and
run/valid/validate_{fiq,cirr,shoes}.py (same function names, positional signatures and return tuples)."""
