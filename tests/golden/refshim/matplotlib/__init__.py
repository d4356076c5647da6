"""stand-in"""
