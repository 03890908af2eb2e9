/* smash_oracle_vic.h -- VIC operators of the vic-a structure, forward and adjoint (included by smash_oracle.c).
 *
 * TEST INFRASTRUCTURE ONLY.  Forward: smash/solver/operator/md_vic_operator.f90:22-202, statement by statement.
 * Adjoint: the local adjoint expressions of forward_db.f90 in their order -- VIC_INFILTRATION_B :6808-6973,
 * VIC_VERTICAL_TRANSFER_B :7102-7197, VIC_INTERFLOW_B :7300-7368, VIC_BASEFLOW_B :7442-7503,
 * BROOKS_AND_COREY_FLOW_B :7582-7643, LINEAR_EVAPOTRANSPIRATION_B :7700-7723 -- in store-all form: each _b routine
 * receives the values its operator saw on entry and recomputes what the Tapenade code pops from its stack. */

static void vic_infiltration(float prcp, float cusl1, float cusl2, float b, float* husl1, float* husl2, float* runoff) {
    float bp1 = b + 1.f, ifl;
    if (prcp <= 0.f) ifl = 0.f;
    else {
        float cusl = cusl1 + cusl2;
        float wusl = (*husl1) * cusl1 + (*husl2) * cusl2;
        wusl = fmaxf(1.e-6f, wusl);
        wusl = fminf(cusl - 1e-6f, wusl);
        float iflm = cusl * bp1;
        float iflc = iflm * (1.f - powf(1.f - (wusl / cusl), 1.f / bp1));
        if (iflc + prcp >= iflm) ifl = cusl - wusl;
        else ifl = (cusl - wusl) - cusl * powf(1.f - ((iflc + prcp) / iflm), bp1);
        ifl = fminf(prcp, ifl);
    }
    float ifl_usl1 = fminf((1.f - *husl1) * cusl1, ifl);
    ifl = ifl - ifl_usl1;
    float ifl_usl2 = fminf((1.f - *husl2) * cusl2, ifl);
    ifl = ifl - ifl_usl2;
    *husl1 = *husl1 + ifl_usl1 / cusl1;
    *husl2 = *husl2 + ifl_usl2 / cusl2;
    *runoff = prcp - (ifl_usl1 + ifl_usl2);
}

static float brooks_and_corey_flow(float ks, float residual, float porosity, float lambda, float c_upper, float c_lower,
                                   float h_upper, float h_lower) {
    float flow = ks * powf((h_upper - residual) / (porosity - residual), lambda);
    float w_upper = h_upper * c_upper * porosity;
    float w_lower = h_lower * c_lower * porosity;
    float max_flow = fminf(w_upper, c_lower - w_lower);
    return fminf(max_flow, flow);
}
static float linear_evapotranspiration(float e, float c, float h) { return fminf(c * h, e * h); }

static void vic_vertical_transfer(float pet, float cusl1, float cusl2, float clsl, float ks, float* husl1, float* husl2, float* hlsl) {
    float fbc = brooks_and_corey_flow(ks, 0.f, 1.f, 1.f, cusl1, cusl2, *husl1, *husl2);
    *husl1 = *husl1 - fbc / cusl1;
    *husl2 = *husl2 + fbc / cusl2;
    fbc = brooks_and_corey_flow(ks, 0.f, 1.f, 1.f, cusl2, clsl, *husl2, *hlsl);
    *husl2 = *husl2 - fbc / cusl2;
    *hlsl = *hlsl + fbc / clsl;
    float fe = linear_evapotranspiration(pet, cusl1, *husl1);
    *husl1 = *husl1 - fe / cusl1;
    float pet_remain = fmaxf(0.f, pet - fe);
    fe = linear_evapotranspiration(pet_remain, cusl2, *husl2);
    *husl2 = *husl2 - fe / cusl2;
    pet_remain = fmaxf(0.f, pet_remain - fe);
    fe = linear_evapotranspiration(pet_remain, clsl, *hlsl);
    *hlsl = *hlsl - fe / clsl;
}

static void vic_interflow(float n, float cusl2, float* husl2, float* qi) {
    float nm1 = n - 1.f, d1pnm1 = 1.f / nm1;
    float husl2_imd = *husl2;
    *husl2 = powf(powf(husl2_imd * cusl2, -nm1) + powf(cusl2, -nm1), -d1pnm1) / cusl2;
    *qi = (husl2_imd - *husl2) * cusl2;
}

static void vic_baseflow(float clsl, float ds, float dsm, float ws, float* hlsl, float* qb) {
    float q;
    if (*hlsl <= ws) q = (ds * dsm) / ws * (*hlsl);
    else q = dsm * (1.f - ds / ws) * (*hlsl - ws) / (1.f - ws);
    float wlsl = clsl * (*hlsl);
    q = fminf(wlsl, q);
    *hlsl = *hlsl - q / clsl;
    *qb = q;
}

/* ---------------------------------------------------------------- adjoints */
static float pow_guard_b(float x, float y, float r_b) {   /* d/dx of x**y times r_b, with the Tapenade guard */
    if (x <= 0.f && (y == 0.f || y != (float)(int)y)) return 0.f;
    return y * powf(x, y - 1.f) * r_b;
}

static void vic_infiltration_b(float prcp, float cusl1, float* cusl1_b, float cusl2, float* cusl2_b, float b, float* b_b,
                               float husl1, float* husl1_b, float husl2, float* husl2_b, float runoff_b) {
    float bp1 = b + 1.f, ifl, cusl = 0.f, wusl = 0.f, iflm = 0.f, iflc = 0.f, pwx1 = 0.f, pwy1 = 0.f, pwr1 = 0.f, pwr1_first = 0.f;
    int c_prcp, c_w1 = 0, c_w2 = 0, c_full = 0, c_min;
    if (prcp <= 0.f) { ifl = 0.f; c_prcp = 0; c_min = 0; }
    else {
        c_prcp = 1;
        cusl = cusl1 + cusl2;
        wusl = husl1 * cusl1 + husl2 * cusl2;
        if (1.e-6f < wusl) c_w1 = 0; else { wusl = 1.e-6f; c_w1 = 1; }
        if (cusl - 1e-6f > wusl) c_w2 = 0; else { wusl = cusl - 1e-6f; c_w2 = 1; }
        iflm = cusl * bp1;
        pwx1 = 1.f - wusl / cusl;
        pwy1 = 1.f / bp1;
        pwr1 = powf(pwx1, pwy1);
        iflc = iflm * (1.f - pwr1);
        if (iflc + prcp >= iflm) { ifl = cusl - wusl; c_full = 1; }
        else {
            pwx1 = 1.f - (iflc + prcp) / iflm;
            pwr1_first = pwr1;
            pwr1 = powf(pwx1, bp1);
            ifl = cusl - wusl - cusl * pwr1;
            c_full = 0;
        }
        if (prcp > ifl) c_min = 2; else { ifl = prcp; c_min = 1; }
    }
    float ifl_usl1, ifl_usl2;
    int c_u1, c_u2;
    if ((1.f - husl1) * cusl1 > ifl) { ifl_usl1 = ifl; c_u1 = 0; } else { ifl_usl1 = (1.f - husl1) * cusl1; c_u1 = 1; }
    ifl = ifl - ifl_usl1;
    if ((1.f - husl2) * cusl2 > ifl) { ifl_usl2 = ifl; c_u2 = 0; } else { ifl_usl2 = (1.f - husl2) * cusl2; c_u2 = 1; }
    float ifl_usl1_b = *husl1_b / cusl1 - runoff_b;
    float ifl_usl2_b = *husl2_b / cusl2 - runoff_b;
    *cusl2_b = *cusl2_b - ifl_usl2 * (*husl2_b) / (cusl2 * cusl2);
    *cusl1_b = *cusl1_b - ifl_usl1 * (*husl1_b) / (cusl1 * cusl1);
    float ifl_b;
    if (c_u2 == 0) ifl_b = ifl_usl2_b;
    else {
        *husl2_b = *husl2_b - cusl2 * ifl_usl2_b;
        *cusl2_b = *cusl2_b + (1.f - husl2) * ifl_usl2_b;
        ifl_b = 0.f;
    }
    ifl_usl1_b = ifl_usl1_b - ifl_b;
    if (c_u1 == 0) ifl_b = ifl_b + ifl_usl1_b;
    else {
        *husl1_b = *husl1_b - cusl1 * ifl_usl1_b;
        *cusl1_b = *cusl1_b + (1.f - husl1) * ifl_usl1_b;
    }
    float bp1_b;
    if (c_prcp == 0) bp1_b = 0.f;
    else {
        if (c_min == 1) ifl_b = 0.f;
        float cusl_b, wusl_b, iflc_b, iflm_b, pwr1_b, pwx1_b, pwy1_b;
        if (c_full == 0) {
            cusl_b = (1.0f - pwr1) * ifl_b;
            wusl_b = -ifl_b;
            pwr1_b = -(cusl * ifl_b);
            pwr1 = pwr1_first;
            if (pwx1 <= 0.0f && (bp1 == 0.0f || bp1 != (float)(int)bp1)) pwx1_b = 0.f;
            else pwx1_b = bp1 * powf(pwx1, bp1 - 1.f) * pwr1_b;
            if (pwx1 <= 0.0f) bp1_b = 0.f;
            else bp1_b = powf(pwx1, bp1) * logf(pwx1) * pwr1_b;
            iflc_b = -(pwx1_b / iflm);
            iflm_b = (prcp + iflc) * pwx1_b / (iflm * iflm);
            pwy1 = 1.f / bp1;
            pwx1 = 1.f - wusl / cusl;
        } else {
            cusl_b = ifl_b;
            wusl_b = -ifl_b;
            iflc_b = 0.f;
            iflm_b = 0.f;
            bp1_b = 0.f;
        }
        iflm_b = iflm_b + (1.f - pwr1) * iflc_b;
        pwr1_b = -(iflm * iflc_b);
        if (pwx1 <= 0.0f && (pwy1 == 0.0f || pwy1 != (float)(int)pwy1)) pwx1_b = 0.f;
        else pwx1_b = pwy1 * powf(pwx1, pwy1 - 1.f) * pwr1_b;
        if (pwx1 <= 0.0f) pwy1_b = 0.f;
        else pwy1_b = powf(pwx1, pwy1) * logf(pwx1) * pwr1_b;
        bp1_b = bp1_b + cusl * iflm_b - pwy1_b / (bp1 * bp1);
        wusl_b = wusl_b - pwx1_b / cusl;
        cusl_b = cusl_b + wusl * pwx1_b / (cusl * cusl) + bp1 * iflm_b;
        if (c_w2 != 0) { cusl_b = cusl_b + wusl_b; wusl_b = 0.f; }
        if (c_w1 != 0) wusl_b = 0.f;
        *husl1_b = *husl1_b + cusl1 * wusl_b;
        *cusl1_b = *cusl1_b + husl1 * wusl_b + cusl_b;
        *husl2_b = *husl2_b + cusl2 * wusl_b;
        *cusl2_b = *cusl2_b + husl2 * wusl_b + cusl_b;
    }
    *b_b = *b_b + bp1_b;
}

static void brooks_and_corey_flow_b(float ks, float* ks_b, float residual, float porosity, float lambda, float c_upper,
                                    float* c_upper_b, float c_lower, float* c_lower_b, float h_upper, float* h_upper_b,
                                    float h_lower, float* h_lower_b, float flow_b) {
    float pwx1 = (h_upper - residual) / (porosity - residual);
    float pwr1 = powf(pwx1, lambda);
    float flow = ks * pwr1;
    float w_upper = h_upper * c_upper * porosity;
    float w_lower = h_lower * c_lower * porosity;
    float max_flow, max_flow_b, w_lower_b, w_upper_b;
    int br;
    if (w_upper > c_lower - w_lower) { max_flow = c_lower - w_lower; br = 0; } else { max_flow = w_upper; br = 1; }
    if (max_flow > flow) max_flow_b = 0.f;
    else { max_flow_b = flow_b; flow_b = 0.f; }
    if (br == 0) { *c_lower_b = *c_lower_b + max_flow_b; w_lower_b = -max_flow_b; w_upper_b = 0.f; }
    else { w_upper_b = max_flow_b; w_lower_b = 0.f; }
    float pwr1_b = ks * flow_b;
    float pwx1_b = pow_guard_b(pwx1, lambda, pwr1_b);
    *h_lower_b = *h_lower_b + c_lower * porosity * w_lower_b;
    *c_lower_b = *c_lower_b + h_lower * porosity * w_lower_b;
    *h_upper_b = *h_upper_b + c_upper * porosity * w_upper_b + pwx1_b / (porosity - residual);
    *c_upper_b = *c_upper_b + h_upper * porosity * w_upper_b;
    *ks_b = *ks_b + pwr1 * flow_b;
}

static void linear_evapotranspiration_b(float e, float* e_b, float c, float* c_b, float h, float* h_b, float flow_b) {
    float flow = e * h, w = c * h, w_b;
    if (w > flow) w_b = 0.f;
    else { w_b = flow_b; flow_b = 0.f; }
    *c_b = *c_b + h * w_b;
    *h_b = *h_b + c * w_b + e * flow_b;
    *e_b = *e_b + h * flow_b;
}

/* husl1, husl2, hlsl: the levels on entry of vic_vertical_transfer */
static void vic_vertical_transfer_b(float pet, float cusl1, float* cusl1_b, float cusl2, float* cusl2_b, float clsl, float* clsl_b,
                                    float ks, float* ks_b, float husl1, float* husl1_b, float husl2, float* husl2_b, float hlsl,
                                    float* hlsl_b) {
    /* forward, keeping what the reverse part needs */
    const float h1_0 = husl1, h2_0 = husl2;
    float fbc1 = brooks_and_corey_flow(ks, 0.f, 1.f, 1.f, cusl1, cusl2, husl1, husl2);
    husl1 = husl1 - fbc1 / cusl1;
    husl2 = husl2 + fbc1 / cusl2;
    const float h2_1 = husl2, hl_1 = hlsl;
    float fbc2 = brooks_and_corey_flow(ks, 0.f, 1.f, 1.f, cusl2, clsl, husl2, hlsl);
    husl2 = husl2 - fbc2 / cusl2;
    hlsl = hlsl + fbc2 / clsl;
    const float h1_2 = husl1, h2_2 = husl2, hl_2 = hlsl;
    float fe1 = linear_evapotranspiration(pet, cusl1, husl1);
    float pet_remain1;
    int br1, br2;
    if (0.f < pet - fe1) { pet_remain1 = pet - fe1; br1 = 0; } else { pet_remain1 = 0.f; br1 = 1; }
    float fe2 = linear_evapotranspiration(pet_remain1, cusl2, h2_2);
    float pet_remain2;
    if (0.f < pet_remain1 - fe2) { pet_remain2 = pet_remain1 - fe2; br2 = 0; } else { pet_remain2 = 0.f; br2 = 1; }
    float fe3 = linear_evapotranspiration(pet_remain2, clsl, hl_2);
    /* reverse */
    float fe_b = -(*hlsl_b / clsl);
    *clsl_b = *clsl_b + fe3 * (*hlsl_b) / (clsl * clsl);
    float pet_remain_b = 0.f;
    linear_evapotranspiration_b(pet_remain2, &pet_remain_b, clsl, clsl_b, hl_2, hlsl_b, fe_b);
    if (br2 == 0) fe_b = -pet_remain_b;
    else { pet_remain_b = 0.f; fe_b = 0.f; }
    fe_b = fe_b - *husl2_b / cusl2;
    *cusl2_b = *cusl2_b + fe2 * (*husl2_b) / (cusl2 * cusl2);
    linear_evapotranspiration_b(pet_remain1, &pet_remain_b, cusl2, cusl2_b, h2_2, husl2_b, fe_b);
    if (br1 == 0) fe_b = -pet_remain_b;
    else fe_b = 0.f;
    fe_b = fe_b - *husl1_b / cusl1;
    *cusl1_b = *cusl1_b + fe1 * (*husl1_b) / (cusl1 * cusl1);
    float pet_b = 0.f;
    linear_evapotranspiration_b(pet, &pet_b, cusl1, cusl1_b, h1_2, husl1_b, fe_b);
    float fbc_b = *hlsl_b / clsl - *husl2_b / cusl2;
    *clsl_b = *clsl_b - fbc2 * (*hlsl_b) / (clsl * clsl);
    *cusl2_b = *cusl2_b + fbc2 * (*husl2_b) / (cusl2 * cusl2);
    brooks_and_corey_flow_b(ks, ks_b, 0.f, 1.f, 1.f, cusl2, cusl2_b, clsl, clsl_b, h2_1, husl2_b, hl_1, hlsl_b, fbc_b);
    fbc_b = *husl2_b / cusl2 - *husl1_b / cusl1;
    *cusl2_b = *cusl2_b - fbc1 * (*husl2_b) / (cusl2 * cusl2);
    *cusl1_b = *cusl1_b + fbc1 * (*husl1_b) / (cusl1 * cusl1);
    brooks_and_corey_flow_b(ks, ks_b, 0.f, 1.f, 1.f, cusl1, cusl1_b, cusl2, cusl2_b, h1_0, husl1_b, h2_0, husl2_b, fbc_b);
}

/* husl2: level on entry */
static void vic_interflow_b(float n, float cusl2, float* cusl2_b, float husl2, float* husl2_b, float qi_b) {
    float nm1 = n - 1.f, d1pnm1 = 1.f / nm1;
    float husl2_imd = husl2;
    float pwx1 = husl2_imd * cusl2, pwy1 = -nm1;
    float pwr1 = powf(pwx1, pwy1);
    float pwy2 = -nm1;
    float pwr2 = powf(cusl2, pwy2);
    float pwx3 = pwr1 + pwr2, pwy3 = -d1pnm1;
    float pwr3 = powf(pwx3, pwy3);
    float husl2_new = pwr3 / cusl2;
    float hb = *husl2_b - cusl2 * qi_b;
    float pwr3_b = hb / cusl2;
    float pwx3_b = pow_guard_b(pwx3, pwy3, pwr3_b);
    float pwr1_b = pwx3_b, pwr2_b = pwx3_b;
    float pwx1_b = pow_guard_b(pwx1, pwy1, pwr1_b);
    float husl2_imd_b = cusl2 * qi_b + cusl2 * pwx1_b;
    if (cusl2 <= 0.0f && (pwy2 == 0.0f || pwy2 != (float)(int)pwy2))
        *cusl2_b = *cusl2_b + (husl2_imd - husl2_new) * qi_b + husl2_imd * pwx1_b - pwr3 * hb / (cusl2 * cusl2);
    else
        *cusl2_b = *cusl2_b + (husl2_imd - husl2_new) * qi_b + pwy2 * powf(cusl2, pwy2 - 1.f) * pwr2_b - pwr3 * hb / (cusl2 * cusl2) +
                   husl2_imd * pwx1_b;
    *husl2_b = husl2_imd_b;
}

/* hlsl: level on entry */
static void vic_baseflow_b(float clsl, float* clsl_b, float ds, float* ds_b, float dsm, float* dsm_b, float ws, float* ws_b, float hlsl,
                           float* hlsl_b, float qb_b) {
    float qb, wlsl;
    int br1, br2;
    if (hlsl <= ws) { qb = ds * dsm / ws * hlsl; br1 = 1; }
    else { qb = dsm * (1.f - ds / ws) * (hlsl - ws) / (1.f - ws); br1 = 0; }
    wlsl = clsl * hlsl;
    if (wlsl > qb) br2 = 0; else { qb = wlsl; br2 = 1; }
    qb_b = qb_b - *hlsl_b / clsl;
    *clsl_b = *clsl_b + qb * (*hlsl_b) / (clsl * clsl);
    float wlsl_b;
    if (br2 == 0) wlsl_b = 0.f;
    else { wlsl_b = qb_b; qb_b = 0.f; }
    *clsl_b = *clsl_b + hlsl * wlsl_b;
    *hlsl_b = *hlsl_b + clsl * wlsl_b;
    if (br1 == 0) {
        float temp = dsm / (-ws + 1.f);
        float temp_b0 = -((hlsl - ws) * temp * qb_b / ws);
        float temp_b1 = (1.f - ds / ws) * qb_b;
        *hlsl_b = *hlsl_b + temp * temp_b1;
        float temp_b = (hlsl - ws) * temp_b1 / (1.f - ws);
        *ws_b = *ws_b + temp * temp_b - temp * temp_b1 - ds * temp_b0 / ws;
        *dsm_b = *dsm_b + temp_b;
        *ds_b = *ds_b + temp_b0;
    } else {
        float temp = hlsl / ws;
        *ds_b = *ds_b + dsm * temp * qb_b;
        *dsm_b = *dsm_b + ds * temp * qb_b;
        float temp_b = ds * dsm * qb_b / ws;
        *hlsl_b = *hlsl_b + temp_b;
        *ws_b = *ws_b - temp * temp_b;
    }
}
