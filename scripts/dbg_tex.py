import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),'tests'))
import pytest
sys.exit(pytest.main(['-q','-m','gpu','tests/test_gpu_parity.py','-k','texture_unit','-x']))
