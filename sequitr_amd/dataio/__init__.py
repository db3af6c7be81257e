"""Data readers kept from the reference's sequitr/dataio (Python-3 restatements of the parts the tile
front end needs)."""
from .octopus import OctopusData  # noqa: F401
