"""TEST INFRASTRUCTURE: empty stand-in so that /root/reference/utils/degradation_utils.py imports (it only names these)."""
