"""Kept for the test modules that import it by this name: the stand-ins live in oracle/engine.py."""
from oracle.engine import OracleEngine, OracleVocoder  # noqa: F401
