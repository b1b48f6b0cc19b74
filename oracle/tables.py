"""Model constants of the oracle -- LITERAL copies, deliberately NOT imported from ``pdb2reaction_amd``.

TEST INFRASTRUCTURE ONLY.  The checker must not share tables with the thing it checks: if the product's
``weights.py`` carried a wrong permutation or epsilon, an oracle importing the same table would agree with it and
every parity test would stay green ("common-mode path").  Each value below is written out from the
specification (the recalled fairchem-core 2.x UMA-S / eSCN-MD hyper-parameters) and
``tests/test_oracle.py::test_oracle_tables_equal_product_tables`` asserts that product and oracle agree -- a
disagreement is then an explicit test failure instead of a silent common mode.
"""
# SURVEY.md App. A header: lmax = mmax = 2 -> (lmax+1)^2 = 9 coefficients; C = H = edge_channels = 128; 4 layers
LMAX = 2
MMAX = 2
NUM_SPH = 9
SPHERE_CHANNELS = 128
HIDDEN_CHANNELS = 128
EDGE_CHANNELS = 128
NUM_LAYERS = 4
# App. A header / A.4: 64 gaussians on [0, cutoff], cutoff 6.0 Angstrom, max_neighbors 300; x_edge = 64 + 2*128 = 320
NUM_DISTANCE_BASIS = 64
CUTOFF = 6.0
MAX_NEIGHBORS = 300
EDGE_FEAT = 320
RADIAL_HIDDEN = 128
MAX_NUM_ELEMENTS = 100
# App. A.6: edge-degree embedding is divided by 5.0 before the scatter
DEG_RESCALE = 5.0
# App. A.5: charge table is indexed by charge + 100 (201 rows), spin by multiplicity (101 rows)
CHARGE_OFFSET = 100
NUM_CHARGE = 201
NUM_SPIN = 101
# App. A header: dataset_list order
DATASET_LIST = ("oc20", "omol", "omat", "odac", "omc")
# App. A.7 RMSNormSH eps; RadialMLP LayerNorm eps (torch.nn.LayerNorm default)
NORM_EPS = 1e-5
LN_EPS = 1e-5
# App. A.3: m-primary row r holds l-primary coefficient TO_M[r], l-primary index = l*l + l + m:
#   m=0: (l0,0)=0 (l1,0)=2 (l2,0)=6 | m=+1: (l1,+1)=3 (l2,+1)=7 ; m=-1: (l1,-1)=1 (l2,-1)=5 | m=+2: (l2,+2)=8 ; m=-2: (l2,-2)=4
TO_M = (0, 2, 6, 3, 7, 1, 5, 8, 4)
# degree of each l-primary coefficient (index l*l+l+m) and of each m-primary row
L_OF_LP = (0, 1, 1, 1, 2, 2, 2, 2, 2)
L_OF_MP = (0, 1, 2, 1, 2, 1, 2, 2, 2)
# App. A.7 / D: parameter shapes of one block (nn.Linear layout [out, in]) and the radial output width
SHAPES = {
    "so2_conv_1.fc_m0.weight": (640, 768),            # (lmax*H gate scalars + 3*H) x (3 * 2C)
    "so2_conv_1.so2_m_conv.0.fc.weight": (512, 512),  # m=1: 2*(2*H) x (2 * 2C)
    "so2_conv_1.so2_m_conv.1.fc.weight": (256, 256),  # m=2: 2*(1*H) x (1 * 2C)
    "so2_conv_1.rad_func.fc3.weight": (1536, 128),    # 768 + 512 + 256 radial weights
    "so2_conv_2.fc_m0.weight": (384, 384),
    "so2_conv_2.so2_m_conv.0.fc.weight": (512, 256),
    "so2_conv_2.so2_m_conv.1.fc.weight": (256, 128),
    "edge_degree_embedding.rad_func.fc3.weight": (384, 128),
    "rad_func.fc1.weight": (128, 320),
    "atom_wise.scalar_mlp.weight": (256, 128),
    "atom_wise.so3_linear_1.weight": (3, 128, 128),
    "energy_block.4.weight": (1, 128),
}
