"""arrowspace_amd: host-side mirror of the arrowspace API over libarrowspace_hip.so (C ABI)."""
