"""Alias package: lets drivers written against the reference (`from tn_gradient.layer.sow import
SoWLinear`, `from tn_gradient.prepare import prepare_sow, ...`) import the MI355X implementation
unchanged.  Everything re-exports from sow_amd."""
