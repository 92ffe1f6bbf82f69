"""`uibk.deep_preconditioning.data_set` (data_set.py:23-214): the folder-backed data sets on the MI355X path."""
from deeppreconditioning_amd.data_set import ROOT, SludgePatternDataSet, StAnDataSet  # noqa: F401
