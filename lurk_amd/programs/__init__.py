"""Lair programs shipped with the harness (synthetic workloads; not part of the reference)."""
