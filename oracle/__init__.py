"""ORACLE package marker.  Test infrastructure only: nothing under lurk_amd/ imports this."""
