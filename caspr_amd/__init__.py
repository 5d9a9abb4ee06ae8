"""caspr_amd -- MI355X-native CaSPR encode -> advect -> sample path (drop-in for caspr.models.CaSPR)."""
__version__ = "0.1.0"
