// TEST INFRASTRUCTURE — NOT PRODUCT CODE, NOT GOOGLETEST.
// The handful of GoogleTest macros the reference's QP solver tests use (TEST, EXPECT_TRUE/EQ/LT/LE/GE/GT/NEAR, InitGoogleTest,
// RUN_ALL_TESTS), so those files can be compiled unchanged against the facade in an image without GoogleTest.
#pragma once
#include <cstdio>
#include <cmath>
#include <exception>
#include <vector>

namespace testing {
struct TestCase { const char *suite, *name; void (*fn)(); };
inline std::vector<TestCase> &registry() { static std::vector<TestCase> r; return r; }
inline int &failures() { static int f = 0; return f; }
struct Registrar { Registrar(const char *s, const char *n, void (*fn)()) { registry().push_back(TestCase{s, n, fn}); } };
inline void InitGoogleTest(int *, char **) {}
inline int run_all() {
    int failed_tests = 0;
    for (auto &t : registry()) {
        const int before = failures();
        printf("[ RUN      ] %s.%s\n", t.suite, t.name);
        try {
            t.fn();
        } catch (const std::exception &e) {
            printf("  exception: %s\n", e.what());
            failures()++;
        }
        const bool ok = failures() == before;
        printf("[ %s ] %s.%s\n", ok ? "      OK" : " FAILED ", t.suite, t.name);
        failed_tests += !ok;
    }
    printf("[==========] %zu tests ran, %d failed\n", registry().size(), failed_tests);
    return failed_tests ? 1 : 0;
}
}  // namespace testing
#define RUN_ALL_TESTS() ::testing::run_all()
#define TEST(suite, name)                                                              \
    static void suite##_##name##_body();                                               \
    static ::testing::Registrar suite##_##name##_reg(#suite, #name, suite##_##name##_body); \
    static void suite##_##name##_body()
#define GTEST_STUB_CHECK(cond, text)                                                   \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            printf("%s:%d: Failure: %s\n", __FILE__, __LINE__, text);                  \
            ::testing::failures()++;                                                   \
        }                                                                              \
    } while (0)
#define EXPECT_TRUE(a) GTEST_STUB_CHECK((a), #a)
#define EXPECT_FALSE(a) GTEST_STUB_CHECK(!(a), "!(" #a ")")
#define EXPECT_EQ(a, b) GTEST_STUB_CHECK((a) == (b), #a " == " #b)
#define EXPECT_NE(a, b) GTEST_STUB_CHECK((a) != (b), #a " != " #b)
#define EXPECT_LT(a, b) GTEST_STUB_CHECK((a) < (b), #a " < " #b)
#define EXPECT_LE(a, b) GTEST_STUB_CHECK((a) <= (b), #a " <= " #b)
#define EXPECT_GT(a, b) GTEST_STUB_CHECK((a) > (b), #a " > " #b)
#define EXPECT_GE(a, b) GTEST_STUB_CHECK((a) >= (b), #a " >= " #b)
#define EXPECT_NEAR(a, b, tol) GTEST_STUB_CHECK(std::fabs((double)(a) - (double)(b)) <= (tol), #a " near " #b)
#define ASSERT_TRUE EXPECT_TRUE
#define ASSERT_EQ EXPECT_EQ
#define ASSERT_LT EXPECT_LT
