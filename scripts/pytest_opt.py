"""Run a pytest selection in-process with library options preset: python scripts/pytest_opt.py name=value ... -- <pytest args>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
from vlsfr_amd import _lib
i = sys.argv.index("--")
for kv in sys.argv[1:i]:
    k, v = kv.split("=")
    _lib.check(_lib.lib().vlsfr_set_option(k.encode(), int(v)))
sys.exit(pytest.main(sys.argv[i + 1:]))
